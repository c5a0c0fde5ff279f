"""The int8 prefilter may only DISCARD rows it can prove irrelevant, so everything hangs on one
inequality:  |canonical(q, d) - approx(q, d)| <= eps8[q]  for every query and every stored row
(csrc/quant.hip, csrc/prep.hip).  Checked here on all pairs of small shards built to stress it:
outlier dimensions, rows and tiles of wildly different magnitude, tiny and huge values, sparse rows,
un-normalised inner-product data.  Also: the bound is not vacuous (eps8 within 3x of the worst error
seen on plain Gaussian data)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _worst_ratio(X, Q, metric):
    import torch

    from rag_arc_amd.hip import binding as B
    from rag_arc_amd.hip.engine import FlatIndexF16

    lib = B.load_library()
    n, d = X.shape
    idx = FlatIndexF16(d, metric=metric, scan="q8")
    idx.add(X)
    with idx._lock:
        idx._workspace()
        q = torch.as_tensor(Q, dtype=torch.float32).cuda().contiguous()
        idx._prep(q)
        out = torch.empty((Q.shape[0], n), dtype=torch.float32, device="cuda")
        B.check(lib.rarc_debug_q8_scores(idx._rows.data_ptr(), n, idx.d_pad, idx._qmeta.data_ptr(),
                                         idx._qbuf["qblock"].data_ptr(), Q.shape[0], out.data_ptr(), 0))
        qb = idx._qbuf["qblock"]
        nqd = 256 * idx.d_pad
        q32 = qb[: nqd * 4].view(torch.float32).view(256, idx.d_pad)[: Q.shape[0]].double()
        eps8 = qb[nqd * 7 + 1024: nqd * 7 + 2048].view(torch.float32)[: Q.shape[0]].double()
        rows = idx.rows.double()                                   # stored fp16 rows, exact in float64
        exact = q32 @ rows.T                                       # float64: within 1e-12 of the real dot product
        errs = (exact - out.double()).abs()
        err = errs.max(dim=1).values
        # per-tile form (what the scan applies): inside 32-row tile t the bound shrinks by hq[q]·(R − R_t), where
        # the tile's R_t rides in the high half of its metadata word as an fp16 rounded up
        hq = qb[nqd * 7 + 3072: nqd * 7 + 4096].view(torch.float32)[: Q.shape[0]].double()
        nt = (n + 31) // 32
        words = idx._qmeta[4: 4 + 2 * nt: 2].contiguous().view(torch.int32)
        rt = ((words >> 16) & 0xffff).to(torch.int16).view(torch.float16).double()
        R = idx._qmeta[0].double()
        assert bool((rt <= R * (1 + 2.0 ** -10) + 1e-12).all()) and float(rt.max()) >= float(R) * 0.999
        bonus = hq[:, None] * (R - rt).clamp(min=0.0)[None, :]                         # [nq][tiles]
        eps_tile = (eps8[:, None] - bonus).repeat_interleave(32, dim=1)[:, :n]
        assert bool((errs <= eps_tile).all()), "per-tile bound violated"
    assert bool((err <= eps8).all()), f"bound violated: worst err/eps8 = {(err / eps8).max().item():.3f}"
    return float((err / eps8).max().item())


def test_bound_holds_and_is_not_vacuous_on_gaussian_data():
    rng = np.random.default_rng(0)
    X = rng.standard_normal((6000, 768)).astype(np.float32)
    Q = rng.standard_normal((64, 768)).astype(np.float32)
    r = _worst_ratio(X, Q, "cosine")
    assert r > 0.02      # Cauchy-Schwarz is ~27x loose on random directions at d=768; far from vacuous


@pytest.mark.parametrize("case", ["outlier_dims", "row_magnitudes", "tile_magnitudes", "sparse", "tiny_huge", "aligned"])
def test_bound_holds_on_adversarial_data(case):
    rng = np.random.default_rng(sum(map(ord, case)))
    n, d, nq = 4000, 384, 32
    X = rng.standard_normal((n, d)).astype(np.float32)
    Q = rng.standard_normal((nq, d)).astype(np.float32)
    metric = "cosine"
    if case == "outlier_dims":                      # a few dimensions 50x the rest, as in real encoders
        X[:, [3, 77, 200]] *= 50.0
        Q[:, [3, 77, 200]] *= 30.0
    elif case == "row_magnitudes":                  # inner product, row norms over 6 decades
        X *= np.exp(rng.uniform(-7, 7, (n, 1))).astype(np.float32)
        metric = "ip"
    elif case == "tile_magnitudes":                 # whole 32-row tiles tiny or huge
        X *= np.repeat(np.exp(rng.uniform(-6, 6, (n // 32, 1))), 32, axis=0).astype(np.float32)
        metric = "ip"
    elif case == "sparse":                          # 95 % zeros
        X *= (rng.random((n, d)) < 0.05)
        Q *= (rng.random((nq, d)) < 0.2)
    elif case == "tiny_huge":                       # fp16 subnormals next to values near the fp16 maximum
        X[::2] *= 1e-6
        X[1::2] *= 6000.0
        metric = "ip"
    elif case == "aligned":                         # queries parallel to rows: errors add up coherently
        Q = X[:nq].copy() + 0.01 * rng.standard_normal((nq, d)).astype(np.float32)
    _worst_ratio(X, Q, metric)
