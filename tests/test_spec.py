"""Pin the numeric primitives of include/pgbart_spec.h (through the oracle's pgbo_* hooks)."""
import ctypes as C

import numpy as np


def _lib(oracle):
    return oracle.lib.lib


def test_philox_known_answers(oracle):
    # Random123 kat_vectors for philox4x32-10
    kats = [
        ((0, 0), (0, 0, 0, 0), (0x6627E8D5, 0xE169C58D, 0xBC57AC4C, 0x9B00DBD8)),
        ((0xFFFFFFFF, 0xFFFFFFFF), (0xFFFFFFFF,) * 4, (0x408F276D, 0x41C83B0E, 0xA20BC7C6, 0x6D5451FD)),
        ((0xA4093822, 0x299F31D0), (0x243F6A88, 0x85A308D3, 0x13198A2E, 0x03707344),
         (0xD16CFE09, 0x94FDCCEB, 0x5001E420, 0x24126EA1)),
    ]
    f = _lib(oracle).pgbo_philox
    f.argtypes = [C.c_uint32] * 6 + [C.c_void_p]
    f.restype = None
    for key, ctr, want in kats:
        out = np.zeros(4, np.uint32)
        f(key[0], key[1], *ctr, out.ctypes.data)
        assert tuple(int(x) for x in out) == want


def test_uniform_range_and_addressing(oracle):
    f = _lib(oracle).pgbo_draw2
    f.argtypes = [C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p]
    f.restype = None
    seen = set()
    for it in range(4):
        for rnd in range(3):
            for part in range(5):
                for purpose in range(1, 6):
                    out = np.zeros(2)
                    f(3415, it, rnd, part, purpose, 0, out.ctypes.data)
                    assert 0.0 <= out[0] < 1.0 and 0.0 <= out[1] < 1.0
                    seen.add((out[0], out[1]))
    assert len(seen) == 4 * 3 * 5 * 5  # every address is its own draw
    a, b = np.zeros(2), np.zeros(2)
    f(3415, 7, 1, 2, 3, 0, a.ctypes.data)
    f(3415, 7, 1, 2, 3, 0, b.ctypes.data)
    assert np.array_equal(a, b)  # pure function of the address


def test_math_against_libm(oracle):
    f = _lib(oracle).pgbo_math
    f.argtypes = [C.c_void_p, C.c_int64] + [C.c_void_p] * 4
    f.restype = None
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-700, 700, 20000), rng.uniform(0, 1, 20000)])
    e, l, s, c = (np.zeros_like(x) for _ in range(4))
    f(x.ctypes.data, x.size, e.ctypes.data, l.ctypes.data, s.ctypes.data, c.ctypes.data)
    assert np.max(np.abs(e / np.exp(x) - 1)) < 1e-15
    pos = x > 1e-300
    ref = np.log(x[pos])
    big = np.abs(ref) > 1e-3
    assert np.max(np.abs(l[pos][big] / ref[big] - 1)) < 2e-15
    u = x[(x >= 0) & (x < 1)]
    su, cu = s[(x >= 0) & (x < 1)], c[(x >= 0) & (x < 1)]
    assert np.max(np.abs(su - np.sin(2 * np.pi * u))) < 2e-15
    assert np.max(np.abs(cu - np.cos(2 * np.pi * u))) < 2e-15
    assert np.max(np.abs(su * su + cu * cu - 1)) < 1e-15


def test_box_muller_is_standard_normal(oracle):
    f = _lib(oracle).pgbo_normal2
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    f.restype = None
    rng = np.random.default_rng(1)
    n = 200_000
    u0, u1 = rng.random(n), rng.random(n)
    z0, z1 = np.zeros(n), np.zeros(n)
    f(u0.ctypes.data, u1.ctypes.data, n, z0.ctypes.data, z1.ctypes.data)
    z = np.concatenate([z0, z1])
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1) < 0.01
    assert abs(np.mean(z ** 3)) < 0.03 and abs(np.mean(z ** 4) - 3) < 0.06
    assert abs(np.corrcoef(z0, z1)[0, 1]) < 0.01
    # u0 = 0 must not produce inf/nan (log(1-u0))
    a, b = np.zeros(1), np.zeros(1)
    f(np.zeros(1).ctypes.data, np.zeros(1).ctypes.data, 1, a.ctypes.data, b.ctypes.data)
    assert np.isfinite(a[0]) and np.isfinite(b[0])


def test_quantisation_is_round_half_even_and_saturates(oracle):
    f = _lib(oracle).pgbo_quant
    f.argtypes = [C.c_double, C.c_double, C.c_void_p]
    f.restype = C.c_int64
    sat = np.zeros(1, np.uint32)
    assert f(2.5, 1.0, sat.ctypes.data) == 2
    assert f(3.5, 1.0, sat.ctypes.data) == 4
    assert f(-2.5, 1.0, sat.ctypes.data) == -2
    assert f(0.1, 1024.0, sat.ctypes.data) == 102
    assert sat[0] == 0
    assert f(1e300, 1.0, sat.ctypes.data) == 2 ** 50
    assert f(-1e300, 1.0, sat.ctypes.data) == -(2 ** 50)
    assert f(float("nan"), 1.0, sat.ctypes.data) == 0
    assert sat[0] == 3
    rng = np.random.default_rng(2)
    xs = rng.normal(0, 100, 2000)
    for x in xs:
        assert f(float(x), 2.0 ** 20, sat.ctypes.data) == int(np.rint(x * 2.0 ** 20))


def test_scales_leave_headroom_for_n_terms(oracle):
    f = _lib(oracle).pgbo_scales
    f.argtypes = [C.c_int64, C.c_int, C.c_void_p]
    f.restype = None
    for n, e in [(1, 0), (500, 5), (100_000, 8), (1_000_000, 3), (2 ** 30, 10)]:
        out = np.zeros(6)
        f(n, e, out.ctypes.data)
        c1, c2, cl = out[:3]
        assert c1 * out[3] == 1 and c2 * out[4] == 1 and cl * out[5] == 1
        assert n * (2.0 ** e) * c1 <= 2.0 ** 62      # n saturated terms fit an int64
        assert n * (2.0 ** (2 * e)) * c2 <= 2.0 ** 62
        assert (2.0 ** e) * c1 <= 2.0 ** 50          # a term stays inside the exact-rounding window


def test_scan64_is_an_inclusive_scan_in_fixed_order(oracle):
    f = _lib(oracle).pgbo_scan64
    f.argtypes = [C.c_void_p]
    f.restype = None
    x = np.arange(1, 65, dtype=np.float64)
    y = x.copy()
    f(y.ctypes.data)
    assert np.array_equal(y, np.cumsum(x))                       # exact on integers
    rng = np.random.default_rng(3)
    x = rng.random(64)
    y = x.copy()
    f(y.ctypes.data)
    np.testing.assert_allclose(y, np.cumsum(x), rtol=1e-14)
    # the association order is part of the contract: row-local Hillis-Steele, then row carries
    t = x.copy()
    for d in (1, 2, 4, 8):
        t = np.array([t[i] + t[i - d] if (i & 15) >= d else t[i] for i in range(64)])
    t[16:32] += t[15]
    t[48:64] += t[47]
    t[32:64] += t[31]
    assert np.array_equal(y, t)


def test_weight_pick_follows_inverse_cdf(oracle):
    f = _lib(oracle).pgbo_pick
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double, C.c_void_p]
    f.restype = C.c_int
    lw = np.zeros(64)
    lw[1:6] = np.log([0.1, 0.2, 0.3, 0.25, 0.15])
    W = np.zeros(64)
    picks = [f(lw.ctypes.data, 1, 5, u, W.ctypes.data) for u in (0.0, 0.05, 0.1, 0.29, 0.31, 0.61, 0.86, 0.999)]
    assert picks == [1, 1, 1, 2, 3, 4, 5, 5]
    np.testing.assert_allclose(W[1:6] / W[5], np.cumsum([0.1, 0.2, 0.3, 0.25, 0.15]), atol=1e-10)
    # overwhelming weight differences (the n=100k regime): never NaN, picks the heavy particle
    lw[1:6] = [-1e6, -3.0, -1e5, -2e6, -1e9]
    assert [f(lw.ctypes.data, 1, 5, u, None) for u in (0.01, 0.5, 0.99)] == [2, 2, 2]


def test_log_ndtr_against_scipy_and_mpmath(oracle):
    """pgb_lphi_t (include/pgbart_spec.h, tables by tools/fit_ll_tables.py): log Phi for the probit likelihood from
    ONE table entry and a degree-8 Horner chain.  The likelihood sums are fixed point, so the bar is ABSOLUTE error:
    < 2e-14 for s >= -9 (1 ulp of log Phi(-9) = -43.6 is 7e-15), < 4 ulp of the result below, down to the lower
    bound of a per-row log-likelihood, -2047, which the table returns exactly from s = -63.875 on."""
    import mpmath as mp
    from scipy.special import log_ndtr

    f = _lib(oracle).pgbo_log_ndtr
    f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    f.restype = None

    def ev(x):
        x = np.ascontiguousarray(x, np.float64)
        out = np.empty_like(x)
        f(x.ctypes.data, x.size, out.ctypes.data)
        return out

    rng = np.random.default_rng(1)
    x = np.concatenate([np.linspace(-63.87, 40, 300001), rng.normal(0, 2, 100000), rng.uniform(-0.2, 0.2, 50000),
                        [0.0, -0.0, 1e-300, -1e-300, 2.0 * 15 / 1 - 1e-9, 30.0, -30.0]])
    out, ref = ev(x), log_ndtr(x)
    err = np.abs(out - ref)
    body = x >= -9
    assert err[body].max() < 2e-14
    assert (err[~body] / np.spacing(np.abs(ref[~body]))).max() <= 4.0
    assert out.max() < 1e-16 and out.min() >= -2047.0       # a log-probability up to the fit error at 0
    lin = out[:300001]
    assert np.all(np.diff(lin) >= -4 * np.spacing(np.abs(lin[1:])) - 2e-14)   # monotone up to the error bound
    # independent of SciPy: mpmath at 40 digits on a subsample
    mp.mp.dps = 40
    xs = np.concatenate([rng.uniform(-63.8, 9, 1500), rng.normal(0, 1.5, 1500)])
    for v, g in zip(xs, ev(xs)):
        r = mp.log(mp.erfc(-mp.mpf(float(v)) / mp.sqrt(2)) / 2)
        e = abs(float(mp.mpf(float(g)) - r))
        assert e < (2e-14 if v >= -9 else 4 * np.spacing(abs(float(r)))), (v, e)
    # the lower bound is reached exactly, the upper tail is exactly 0, infinities give the limits
    far = np.array([-63.875, -63.9, -64.0, -100.0, -1e300, -np.inf, 63.9, 64.0, 1e300, np.inf])
    assert np.array_equal(ev(far), [-2047.0] * 6 + [0.0] * 4)
    assert ev([-63.874])[0] > -2047.0
    # the same table through the likelihood entry point: no clamp needed, y only picks the sign
    g = _lib(oracle).pgbo_loglik
    g.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p]
    g.restype = None
    mu = rng.normal(0, 3, 20000)
    y = (rng.uniform(size=mu.size) < 0.5).astype(np.float64)
    ll = np.empty_like(mu)
    g(1, y.ctypes.data, mu.ctypes.data, mu.size, ll.ctypes.data)
    assert np.array_equal(ll, ev(np.where(y > 0.5, mu, -mu)))


def test_table_driven_exp_and_log_of_the_likelihood_pass(oracle):
    """pgb_exp_t / pgb_log_t: exp as 2^k T[j] e^r (32-entry table, degree-6 polynomial), log as k ln2 + log c +
    log1p(z / c - 1) (128-entry table, no division).  exp: < 1 ulp against mpmath; log: <= 2 ulp against NumPy
    everywhere (full relative accuracy around 1), absolute error < 2e-16 max(1, |log x|)."""
    import mpmath as mp

    f = _lib(oracle).pgbo_math_t
    f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]
    f.restype = None

    def ev(x):
        x = np.ascontiguousarray(x, np.float64)
        e, l = np.empty_like(x), np.empty_like(x)
        f(x.ctypes.data, x.size, e.ctypes.data, l.ctypes.data)
        return e, l

    rng = np.random.default_rng(2)
    x = np.concatenate([rng.uniform(-700, 700, 200000), rng.uniform(-40, 40, 200000), rng.normal(0, 1, 100000)])
    e, _ = ev(x)
    ref = np.exp(x)
    assert (np.abs(e - ref) / ref).max() < 1.2 * 2.2204e-16   # NumPy's exp is itself good to ~0.5 ulp
    mp.mp.dps = 40
    xs = rng.uniform(-700, 700, 2000)
    for v, g in zip(xs, ev(xs)[0]):
        r = mp.exp(mp.mpf(float(v)))
        assert abs(float((mp.mpf(float(g)) - r) / r)) < 2.2204e-16, v
    edge = np.array([0.0, -0.0, 709.7, -745.1, -800.0, 800.0, 1e5, -1e5])
    ee = ev(edge)[0]
    assert ee[0] == 1.0 and ee[1] == 1.0 and np.isclose(ee[2], np.exp(709.7), rtol=1e-13)
    assert ee[3] == 5e-324 and ee[4] == 0.0 and ee[5] == np.inf and ee[6] == np.inf and ee[7] == 0.0
    assert np.isnan(ev([np.nan])[0][0])
    y = np.concatenate([np.exp(rng.uniform(-700, 700, 200000)), rng.uniform(0.5, 2, 300000),
                        1 + rng.normal(0, 1e-3, 100000), 1 + rng.normal(0, 1e-6, 100000), rng.uniform(1, 9, 100000)])
    _, l = ev(y)
    ref = np.log(y)
    assert (np.abs(l - ref) / np.spacing(np.abs(ref) + 1e-300)).max() <= 2.0
    sp = np.array([1.0, 2.0, 0.5, 4e-320, 2.2250738585072014e-308, 1.7976931348623157e308, np.inf, 0.0, -1.0])
    ls = ev(sp)[1]
    assert ls[0] == 0.0 and ls[1] == np.log(2.0) and ls[2] == -np.log(2.0)
    assert np.allclose(ls[3:6], np.log(sp[3:6]), rtol=1e-15) and ls[6] == np.inf and ls[7] == -1e300 and ls[8] == -1e300
    assert np.isnan(ev([np.nan])[1][0])


def test_log_ndtr_local_variable_forms_agree_bit_for_bit():
    """`pgb_lphi_t` (include/pgbart_spec.h) builds its local variable u = 16 t - 1 from the mantissa bits below
    the 3 interval bits as fma(2, 1 + 8 t, -3); the textbook form is (m - (1 + sub/8)) * 16 - 1 with m the mantissa
    in [1, 2).  Every intermediate of either form is exactly representable, so they must agree to the last bit --
    checked here on 2 million random bit patterns and the interval edges (NumPy: 2 * m8 is exact, so 2 * m8 - 3 is
    what the fma returns)."""
    rng = np.random.default_rng(7)
    zb = rng.integers(0, 2**63 - 1, size=2_000_000, dtype=np.int64).astype(np.uint64)
    edges = np.array([0x3FF0000000000000, 0x3FF1FFFFFFFFFFFF, 0x3FF2000000000000, 0x3FFFFFFFFFFFFFFF,
                      0x4000000000000001, 0x3FEE000000000000, 0x3FFE000000000001], dtype=np.uint64)
    zb = np.concatenate([zb, edges])
    mant = zb & np.uint64(0x000FFFFFFFFFFFFF)
    sub = ((zb >> np.uint64(49)) & np.uint64(7)).astype(np.float64)
    m = (mant | np.uint64(0x3FF0000000000000)).view(np.float64)
    textbook = (m - (1.0 + sub * 0.125)) * 16.0 - 1.0
    m8 = (((zb & np.uint64(0x0001FFFFFFFFFFFF)) << np.uint64(3)) | np.uint64(0x3FF0000000000000)).view(np.float64)
    fused = 2.0 * m8 - 3.0
    assert np.array_equal(textbook.view(np.uint64), fused.view(np.uint64))
    assert fused.min() >= -1.0 and fused.max() < 1.0
