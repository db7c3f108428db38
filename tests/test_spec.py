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


def test_log_ndtr_against_scipy(oracle):
    # pgb_log_ndtr: table-driven scaled-tail form (tools/fit_log_ndtr.py); the likelihood sums
    # are fixed point, so the bar is absolute error (1 ulp of log Phi(-38) = -726 is 1.1e-13)
    from scipy.special import log_ndtr

    f = _lib(oracle).pgbo_log_ndtr
    f.argtypes = [C.c_void_p, C.c_int64, C.c_void_p]
    f.restype = None
    rng = np.random.default_rng(1)
    x = np.concatenate([np.linspace(-38, 38, 100001), rng.normal(0, 2, 50000),
                        [0.0, -0.0, 1e-300, -1e-300, 2.0 * 15 / 1 - 1e-9, 30.0, -30.0]])
    out = np.empty_like(x)
    f(x.ctypes.data, x.size, out.ctypes.data)
    ref = log_ndtr(x)
    assert np.max(np.abs(out - ref)) < 4e-13
    body = np.abs(x) < 5
    assert np.max(np.abs(out[body] - ref[body])) < 1e-14  # 3 ulp of 15
    assert np.all(out <= 0.0) and np.all(np.diff(out[:100001]) >= -1e-15)  # a log-probability, monotone
    # far tails stay finite and follow -x^2/2 - log(|x| sqrt(2 pi))
    far = np.array([-1e3, -1e5, -1e8, 1e3, 1e8])
    o2 = np.empty_like(far)
    f(far.ctypes.data, far.size, o2.ctypes.data)
    assert np.allclose(o2[:3], log_ndtr(far[:3]), rtol=1e-14) and np.all(o2[3:] == 0.0)
    nan = np.array([np.nan])
    f(nan.ctypes.data, 1, nan.ctypes.data)
    assert np.isnan(nan[0])


def test_log_ndtr_local_variable_forms_agree_bit_for_bit():
    """`pgb_log_ndtr_t` (include/pgbart_spec.h) builds its local variable u = 16 t - 1 from the mantissa bits below
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
