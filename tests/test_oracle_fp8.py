"""CPU: the oracle's e4m3fn codec (oracle/rarc_oracle.c: f8_encode / f8_decode) against the format's
definition — every code round-trips, codes are monotone, ties round to even, saturation at 448 — and the
fp8 ingest / search against a float64 restatement."""
import numpy as np


def _codes(oracle):
    codes = np.arange(256, dtype=np.uint8)
    return codes, oracle.f8_decode(codes)


def test_decode_matches_the_format_definition(oracle):
    codes, vals = _codes(oracle)
    for c in range(256):
        s, e, m = c >> 7, (c >> 3) & 15, c & 7
        if e == 15 and m == 7:
            continue                                  # the NaN code
        want = (m * 2.0 ** -9) if e == 0 else (1 + m / 8) * 2.0 ** (e - 7)
        assert vals[c] == (-want if s else want)
    assert vals[0x7e] == 448.0 and vals[0x08] == 2.0 ** -6 and vals[0x01] == 2.0 ** -9


def test_encode_round_trips_rounds_to_even_and_saturates(oracle):
    codes, vals = _codes(oracle)
    ok = np.array([c for c in codes if (c & 0x7f) != 0x7f])
    re = oracle.f8_encode(vals[ok])
    assert np.array_equal(re[ok != 0x80], ok[ok != 0x80])          # -0 encodes with its sign bit: accept both
    pos = vals[:0x7f].astype(np.float64)
    mids = (pos[:-1] + pos[1:]) / 2                                 # exact ties between neighbours
    enc = oracle.f8_encode(mids.astype(np.float32))
    assert all(e in (i, i + 1) and e % 2 == 0 for i, e in enumerate(enc))   # tie -> even code
    just_above = np.nextafter(mids.astype(np.float32), np.float32(np.inf))
    assert np.array_equal(oracle.f8_encode(just_above), np.arange(1, 0x7f))
    assert list(oracle.f8_encode(np.array([448.0, 449.0, 1e9, np.inf, -1e9], np.float32))) == [0x7e, 0x7e, 0x7e, 0x7e, 0xfe]


def test_ingest_and_search_against_float64(oracle):
    rng = np.random.default_rng(3)
    X = rng.standard_normal((700, 200)).astype(np.float32)
    Q = rng.standard_normal((6, 200)).astype(np.float32)
    b, s, n2 = oracle.ingest_f8(X)
    assert b.shape == (700, 256) and not b[:, 200:].any()
    Xn = oracle.normalize_L2(X)
    assert np.allclose(s, np.abs(Xn).max(axis=1) / 448.0, rtol=1e-6)
    dec = oracle.f8_decode(b).astype(np.float64) * s[:, None]
    assert np.abs(dec[:, :200] - Xn).max() <= np.abs(Xn).max(axis=1).max() * 2.0 ** -4   # 3 mantissa bits
    assert np.allclose(n2, (dec ** 2).sum(axis=1), rtol=1e-5)
    ids, sc, _ = oracle.flat_search_f8(b, s, oracle.normalize_L2(Q), 10)
    qp = np.zeros((6, 256)); qp[:, :200] = oracle.normalize_L2(Q)
    S = qp @ dec.T
    want = np.argsort(-S, axis=1, kind="stable")[:, :10]
    assert np.array_equal(ids, want)
    assert np.allclose(sc, np.take_along_axis(S, want, axis=1), atol=1e-6)
