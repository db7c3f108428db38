"""The DEVICE compile of include/pgbart_spec.h, looked at by itself.

Both backends compile the numeric contract from one header, so `HIP == oracle` on sampler outputs is
tautological for that layer, and tests/test_spec_independent.py pins only gcc's compile of it.  Here the
functions as hipcc compiled them for gfx950 run on the GPU through the `pgbh_*` probes
(pymc_bart_amd/csrc/pgb_probe.h) and are compared

* with SciPy / NumPy -- the SAME checks as tests/test_spec_independent.py, run against the device, and
* bit for bit with the host compile (`pgbo_*` of the oracle library): another compiler, another FMA
  contraction policy, another libm if one leaked in,

on dense random inputs plus the edge values (0, +-inf, NaN, subnormals, clamp boundaries).
"""
import ctypes as C

import numpy as np
import pytest

import test_spec_independent as T
from pymc_bart_amd import _abi

pytestmark = pytest.mark.gpu

_SAME_SIGNATURE = {  # oracle hook -> device probe with the identical argument list
    "pgbo_loglikq": "pgbh_loglikq",
    "pgbo_loglik_multi": "pgbh_loglik_multi",
    "pgbo_loglik_cat_f": "pgbh_loglik_cat_f",
    "pgbo_log_ndtr": "pgbh_log_ndtr",
    "pgbo_math": "pgbh_math",
    "pgbo_normal2": "pgbh_normal2",
}


class _DeviceSpec:
    """Looks like the `oracle` fixture to the checks of test_spec_independent, answers from the GPU."""

    class _Lib:
        def __init__(self, cdll):
            self._cdll = cdll

        def __getattr__(self, name):
            return getattr(self._cdll, _SAME_SIGNATURE[name])

    class _Outer:
        pass

    def __init__(self, hip):
        self.lib = self._Outer()
        self.lib.lib = self._Lib(hip.lib.lib)


@pytest.mark.parametrize("check", [
    T.test_bernoulli_links_against_scipy,
    T.test_asymmetric_laplace_is_the_quantile_regression_density,
    T.test_student_t_against_scipy,
    T.test_gamma_log_link_against_scipy,
    T.test_count_families_against_scipy,
    T.test_multi_output_families_against_scipy,
    T.test_factorised_softmax_of_constant_leaves_against_scipy,
], ids=lambda f: f.__name__.replace("test_", ""))
def test_device_compile_against_scipy(hip, check):
    check(_DeviceSpec(hip))


# ------------------------------------------------------------------------------------------------
def _fn(lib, name, *argtypes, restype=C.c_int):
    f = getattr(lib, name)
    f.restype, f.argtypes = restype, list(argtypes)
    return f


def _bits(a):
    return np.ascontiguousarray(a, float).view(np.uint64)


def _same_bits(dev, host, what):
    d, h = _bits(dev), _bits(host)
    both_nan = np.isnan(dev) & np.isnan(host)  # (NaN payloads are not part of the contract)
    bad = (d != h) & ~both_nan
    assert not bad.any(), f"{what}: {int(bad.sum())} of {bad.size} differ, first at {int(np.argmax(bad))}: " \
                          f"device {np.asarray(dev).ravel()[np.argmax(bad)]!r} host {np.asarray(host).ravel()[np.argmax(bad)]!r}"


EDGE = np.array([0.0, -0.0, 1.0, -1.0, 0.5, -0.5, 1e-300, -1e-300, 5e-324, 1e300, -1e300, np.inf, -np.inf, np.nan,
                 37.5, -37.5, 38.5, -38.5, 8.0, -8.0, 708.0, -708.0, 709.8, -745.2, 2047.0, -2047.0, 2.0 ** 52, 2.0 ** -52])


def _xs(rng, n, scale):
    x = np.concatenate([rng.normal(0, scale, n), rng.uniform(-40, 40, n), EDGE])
    return np.ascontiguousarray(x)


def test_elementary_functions_device_equals_host(hip, oracle):
    dl, ol = hip.lib.lib, oracle.lib.lib
    rng = np.random.default_rng(101)
    x = np.concatenate([_xs(rng, 150_000, 3.0), rng.uniform(0, 1, 50_000), np.exp(rng.uniform(-700, 700, 50_000))])
    n = x.size
    sig = (C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
    dev, host = [np.zeros(n) for _ in range(4)], [np.zeros(n) for _ in range(4)]
    assert _fn(dl, "pgbh_math", *sig)(x.ctypes.data, n, *[a.ctypes.data for a in dev]) == 0
    _fn(ol, "pgbo_math", *sig, restype=None)(x.ctypes.data, n, *[a.ctypes.data for a in host])
    # (pgb_sincos2pi is defined on [0, 1): its octant index is a double -> int conversion, which C leaves
    #  undefined out of the int range -- x86 and gfx950 differ there -- so it is compared where it converts)
    conv = np.abs(x) < 1e8
    for name, d, h in zip(("exp", "log", "sin2pi", "cos2pi"), dev, host):
        sel = conv if name.endswith("2pi") else slice(None)
        _same_bits(d[sel], h[sel], "pgb_" + name)
    # and against libm, as the host-side test does (the contract's own accuracy claim)
    fin = np.isfinite(x) & (np.abs(x) < 700)
    assert np.max(np.abs(dev[0][fin] / np.exp(x[fin]) - 1.0)) < 1e-15
    pos = np.isfinite(x) & (x > 1e-300)
    assert np.max(np.abs(dev[1][pos] - np.log(x[pos])) / (1.0 + np.abs(np.log(x[pos])))) < 2e-15
    # the table-driven exp / log of the per-row likelihoods, tables staged in LDS as the likelihood pass does
    sig = (C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)
    devt, hostt = [np.zeros(n) for _ in range(2)], [np.zeros(n) for _ in range(2)]
    assert _fn(dl, "pgbh_math_t", *sig)(x.ctypes.data, n, *[a.ctypes.data for a in devt]) == 0
    _fn(ol, "pgbo_math_t", *sig, restype=None)(x.ctypes.data, n, *[a.ctypes.data for a in hostt])
    big = np.abs(x) < 4e7   # (pgb_exp_t reads its integer part from 32 bits: defined up to |x| < 4.6e7)
    _same_bits(devt[0][big], hostt[0][big], "pgb_exp_t")
    _same_bits(devt[1], hostt[1], "pgb_log_t")
    assert np.max(np.abs(devt[0][fin] / np.exp(x[fin]) - 1.0)) < 3e-16
    assert np.max(np.abs(devt[1][pos] - np.log(x[pos])) / np.spacing(np.abs(np.log(x[pos])) + 1e-300)) <= 2.0
    # log Phi
    sig = (C.c_void_p, C.c_int64, C.c_void_p)
    d, h = np.zeros(n), np.zeros(n)
    assert _fn(dl, "pgbh_log_ndtr", *sig)(x.ctypes.data, n, d.ctypes.data) == 0
    _fn(ol, "pgbo_log_ndtr", *sig, restype=None)(x.ctypes.data, n, h.ctypes.data)
    _same_bits(d, h, "pgb_log_ndtr")
    # Box-Muller
    u0, u1 = rng.random(200_000), rng.random(200_000)
    u0[:4] = [2.0 ** -53, 1.0 - 2.0 ** -53, 0.5, 2.0 ** -53]
    u1[:4] = [0.0, 1.0 - 2.0 ** -53, 0.25, 0.75]
    sig = (C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p)
    dz, hz = [np.zeros(u0.size) for _ in range(2)], [np.zeros(u0.size) for _ in range(2)]
    assert _fn(dl, "pgbh_normal2", *sig)(u0.ctypes.data, u1.ctypes.data, u0.size, dz[0].ctypes.data, dz[1].ctypes.data) == 0
    _fn(ol, "pgbo_normal2", *sig, restype=None)(u0.ctypes.data, u1.ctypes.data, u0.size, hz[0].ctypes.data, hz[1].ctypes.data)
    _same_bits(dz[0], hz[0], "pgb_normal2 z0")
    _same_bits(dz[1], hz[1], "pgb_normal2 z1")
    z = np.concatenate(dz)
    assert abs(z.mean()) < 0.01 and abs(z.std() - 1.0) < 0.01


PARAMS = {  # family -> (param, param2) settings
    "bernoulli_probit": [(0.0, 1.0)], "bernoulli_logit": [(0.0, 1.0)], "poisson_log": [(0.0, 1.0)],
    "negbin_log": [(0.5, 1.0), (7.0, 1.0)], "asymmetric_laplace": [(0.25, 0.9), (2.0, 0.1)],
    "student_t": [(0.2, 3.0), (2.5, 30.0)], "gamma_log": [(0.7, 1.0), (12.0, 1.0)],
}


@pytest.mark.parametrize("family", sorted(PARAMS))
def test_per_row_loglikelihood_device_equals_host(hip, oracle, family):
    dl, ol = hip.lib.lib, oracle.lib.lib
    rng = np.random.default_rng(sum(map(ord, family)))
    mu = _xs(rng, 100_000, 2.5)
    n = mu.size
    if family.startswith("bernoulli"):
        y = (rng.random(n) < 0.5).astype(float)
    elif family in ("poisson_log", "negbin_log"):
        y = rng.poisson(3.0, n).astype(float)
    elif family == "gamma_log":
        y = rng.gamma(2.0, 1.5, n)
    else:
        y = rng.normal(0, 2, n)
    y[-EDGE.size:] = np.resize(y[:7], EDGE.size)
    sig = (C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_void_p)
    for param, param2 in PARAMS[family]:
        d, h = np.zeros(n), np.zeros(n)
        assert _fn(dl, "pgbh_loglikq", *sig)(_abi.FAMILIES[family], y.ctypes.data, mu.ctypes.data, n, param, param2,
                                             d.ctypes.data) == 0
        _fn(ol, "pgbo_loglikq", *sig, restype=None)(_abi.FAMILIES[family], y.ctypes.data, mu.ctypes.data, n, param,
                                                    param2, h.ctypes.data)
        _same_bits(d, h, f"pgb_loglik1q[{family}, {param}, {param2}]")
        ok = np.isfinite(mu)  # (a predictor is finite: the boundary refuses non-finite responses / offsets)
        assert np.all((d[ok] <= 1e-16) & (d[ok] >= -2047.0))  # (log Phi from its table: 0 up to the fit error)
        if family.startswith("bernoulli"):
            # the form the likelihood pass runs: sign mask on the predictor, log-Phi tables staged in LDS
            s = np.zeros(n)
            assert _fn(dl, "pgbh_loglik_bern_lds", C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)(
                _abi.FAMILIES[family], y.ctypes.data, mu.ctypes.data, n, s.ctypes.data) == 0
            _same_bits(s, h, f"pgb_loglik_bern_s[{family}] with LDS tables")


def test_multi_output_loglikelihood_device_equals_host(hip, oracle):
    dl, ol = hip.lib.lib, oracle.lib.lib
    rng = np.random.default_rng(103)
    sig = (C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
    n = 60_000
    for fam, Ks in (("categorical", (2, 3, 4, 5, 8)), ("normal_meanscale", (2,))):
        for K in Ks:
            mu = np.ascontiguousarray(rng.normal(0, 3, (n, K)))
            mu[:EDGE.size, 0] = EDGE
            mu[:EDGE.size, K - 1] = EDGE[::-1]
            y = rng.integers(0, K, n).astype(float) if fam == "categorical" else rng.normal(0, 1, n)
            d, h = np.zeros(n), np.zeros(n)
            assert _fn(dl, "pgbh_loglik_multi", *sig)(_abi.FAMILIES[fam], K, y.ctypes.data, mu.ctypes.data, n, d.ctypes.data) == 0
            _fn(ol, "pgbo_loglik_multi", *sig, restype=None)(_abi.FAMILIES[fam], K, y.ctypes.data, mu.ctypes.data, n, h.ctypes.data)
            _same_bits(d, h, f"pgb_loglik[{fam}, K={K}]")
            # the form k_loglik<K> runs: every table in LDS
            dl_ = np.zeros(n)
            assert _fn(dl, "pgbh_loglik_multi_lds", C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_double,
                       C.c_double, C.c_void_p)(_abi.FAMILIES[fam], K, y.ctypes.data, mu.ctypes.data, n, 0.0, 1.0,
                                               dl_.ctypes.data) == 0
            _same_bits(dl_, h, f"pgb_loglikq_t[{fam}, K={K}] with LDS tables")


def test_factorised_softmax_device_equals_host(hip, oracle):
    """pgb_loglik_cat_f (the softmax of constant K-vector leaves as k_loglik<K, categorical> evaluates it): the
    device compile against the host compile, bit for bit, over ordinary predictors, the edge values and the regime
    where the factorised sum is lost and the unfactorised form takes over."""
    dl, ol = hip.lib.lib, oracle.lib.lib
    rng = np.random.default_rng(105)
    sig = (C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
    n = 60_000
    for K in (2, 3, 4, 5, 8, 16):
        for se, sv in ((3.0, 1.0), (300.0, 300.0)):
            eta = np.ascontiguousarray(rng.normal(0, se, (n, K)))
            eta[:EDGE.size, 0] = np.clip(EDGE, -1e6, 1e6)
            eta[:EDGE.size, K - 1] = np.clip(EDGE[::-1], -1e6, 1e6)
            v = rng.normal(0, sv, K)
            y = rng.integers(0, K, n).astype(float)
            d, h = np.zeros(n), np.zeros(n)
            assert _fn(dl, "pgbh_loglik_cat_f", *sig)(K, y.ctypes.data, eta.ctypes.data, v.ctypes.data, n, d.ctypes.data) == 0
            _fn(ol, "pgbo_loglik_cat_f", *sig, restype=None)(K, y.ctypes.data, eta.ctypes.data, v.ctypes.data, n, h.ctypes.data)
            _same_bits(d, h, f"pgb_loglik_cat_f[K={K}, spread {se}/{sv}]")


def test_random_stream_and_fixed_point_device_equals_host(hip, oracle):
    dl, ol = hip.lib.lib, oracle.lib.lib
    n = 4096
    draw_h = _fn(ol, "pgbo_draw2", C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_void_p,
                 restype=None)
    draw_d = _fn(dl, "pgbh_draw2", C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int64, C.c_void_p,
                 C.c_void_p)
    for seed, it, rd, part, purpose in ((0, 0, 0, 0, 1), (2 ** 64 - 1, 2 ** 32 - 1, 255, 63, 6), (123456789, 77, 3, 17, 4)):
        u0, u1 = np.zeros(n), np.zeros(n)
        assert draw_d(seed, it, rd, part, purpose, n, u0.ctypes.data, u1.ctypes.data) == 0
        o = np.zeros(2)
        for sub in (0, 1, 2, 100, n - 1):
            draw_h(seed, it, rd, part, purpose, sub, o.ctypes.data)
            assert (u0[sub], u1[sub]) == (o[0], o[1])
        assert np.all((u0 > 0) & (u0 < 1) & (u1 >= 0) & (u1 < 1))
    # pgb_quant: round-half-even, saturation, NaN -> 0
    rng = np.random.default_rng(104)
    quant_h = _fn(ol, "pgbo_quant", C.c_double, C.c_double, C.c_void_p, restype=C.c_int64)
    x = np.concatenate([rng.normal(0, 50, 20_000), (rng.integers(-10 ** 6, 10 ** 6, 2000) + 0.5) / 2.0 ** 20, EDGE])
    for scale in (2.0 ** 20, 2.0 ** 33, 2.0 ** 45):
        q, sat = np.zeros(x.size, np.int64), np.zeros(x.size, np.uint32)
        assert _fn(dl, "pgbh_quant", C.c_void_p, C.c_int64, C.c_double, C.c_void_p, C.c_void_p)(
            x.ctypes.data, x.size, scale, q.ctypes.data, sat.ctypes.data) == 0
        s1 = C.c_uint32(0)
        for i in list(range(0, 20_000, 97)) + list(range(20_000, x.size)):
            s1.value = 0
            assert quant_h(float(x[i]), scale, C.byref(s1)) == q[i] and s1.value == sat[i], (x[i], scale)
        with np.errstate(over="ignore", invalid="ignore"):
            inside = np.isfinite(x) & (np.abs(x * scale) < 2.0 ** 50)
        assert np.array_equal(q[inside], np.rint(x[inside] * scale).astype(np.int64)) and not sat[inside].any()
    # split rules
    go_h = _fn(ol, "pgbo_go_left", C.c_int, C.c_double, C.c_double, restype=C.c_int)
    xs = np.concatenate([rng.integers(0, 52, 3000).astype(float), EDGE])
    vs = np.concatenate([rng.integers(1, 2 ** 52, 3000).astype(float), EDGE[::-1]])
    for rule in (_abi.RULE_CONTINUOUS, _abi.RULE_ONEHOT, _abi.RULE_SUBSET):
        if rule == _abi.RULE_SUBSET:  # contract: x is not NaN, the split value is a 52-bit mask
            xs = np.where(np.isnan(xs), 3.0, xs)
            vs = np.concatenate([vs[:3000], rng.integers(1, 2 ** 52, EDGE.size).astype(float)])
        out = np.zeros(xs.size, np.int64)
        assert _fn(dl, "pgbh_go_left", C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)(
            rule, xs.ctypes.data, vs.ctypes.data, xs.size, out.ctypes.data) == 0
        want = np.array([go_h(rule, float(a), float(b)) for a, b in zip(xs, vs)])
        assert np.array_equal(out, want), rule


def test_leaf_algebra_device_equals_host_and_numpy(hip, oracle):
    dl, ol = hip.lib.lib, oracle.lib.lib
    sc = T._scales(oracle, 100_000, 6)
    rng = np.random.default_rng(105)
    n = 3000
    cnt = rng.integers(0, 5000, n).astype(np.int64)
    cnt[:4] = [0, 1, 2, 3]
    q_st = (rng.normal(0, 1, n) * cnt * sc["c1"]).astype(np.int64)
    q_r = (rng.normal(0, 1, n) * cnt * sc["c1"]).astype(np.int64)
    q_r2 = (rng.gamma(2, 1, n) * cnt * sc["c2"]).astype(np.int64)
    z = rng.normal(0, 1, n)
    m, sd = 50.0, 0.8
    val, sse = np.zeros(n), np.zeros(n)
    I, D = C.c_void_p, C.c_double
    assert _fn(dl, "pgbh_leaf", I, I, I, I, I, C.c_int64, D, D, D, D, I, I)(
        cnt.ctypes.data, q_st.ctypes.data, q_r.ctypes.data, q_r2.ctypes.data, z.ctypes.data, n, sc["inv_c1"], sc["inv_c2"],
        m, sd, val.ctypes.data, sse.ctypes.data) == 0
    leaf_val = _fn(ol, "pgbo_leaf_value", C.c_int64, C.c_int64, D, D, D, D, restype=D)
    leaf_sse = _fn(ol, "pgbo_leaf_sse", C.c_int64, C.c_int64, C.c_int64, D, D, D, restype=D)
    hv = np.array([leaf_val(int(c), int(a), sc["inv_c1"], m, float(zz), sd) for c, a, zz in zip(cnt, q_st, z)])
    hs = np.array([leaf_sse(int(c), int(a), int(b), float(v), sc["inv_c1"], sc["inv_c2"]) for c, a, b, v in zip(cnt, q_r, q_r2, hv)])
    _same_bits(val, hv, "pgb_leaf_value")
    _same_bits(sse, hs, "pgb_leaf_sse")
    nz = cnt > 0
    assert np.allclose(val[nz], q_st[nz] * sc["inv_c1"] / cnt[nz] / m + z[nz] * sd, rtol=1e-12, atol=1e-12)
    # linear leaves
    range_exp = 6
    R, inv_R = 2.0 ** (range_exp - 1), 2.0 ** (1 - range_exp)
    q_u = (rng.normal(0, 0.3, n) * cnt * R * sc["c1"]).astype(np.int64)
    q_uu = (rng.gamma(2, 0.2, n) * cnt * R * sc["c1"]).astype(np.int64)
    q_us = (rng.normal(0, 1, n) * cnt * sc["c1"]).astype(np.int64)
    q_ur = (rng.normal(0, 1, n) * cnt * sc["c1"]).astype(np.int64)
    sse_c = rng.gamma(2, 50, n)
    outs = [np.zeros(n) for _ in range(4)]
    assert _fn(dl, "pgbh_lin", I, I, I, I, I, I, I, I, C.c_int64, D, D, D, I, I, I, I)(
        cnt.ctypes.data, q_u.ctypes.data, q_uu.ctypes.data, q_us.ctypes.data, q_st.ctypes.data, q_ur.ctypes.data,
        q_r.ctypes.data, sse_c.ctypes.data, n, sc["inv_c1"], inv_R, m, *[o.ctypes.data for o in outs]) == 0
    lin_fit = _fn(ol, "pgbo_lin_fit", C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, D, D, D, I, restype=None)
    lin_sse = _fn(ol, "pgbo_lin_sse", D, D, D, D, C.c_int64, C.c_int64, D, restype=D)
    host = np.zeros((n, 4))
    o3 = np.zeros(3)
    for i in range(n):
        lin_fit(int(cnt[i]), int(q_u[i]), int(q_uu[i]), int(q_us[i]), int(q_st[i]), sc["inv_c1"], inv_R, m, o3.ctypes.data)
        host[i, :3] = o3
        host[i, 3] = lin_sse(float(sse_c[i]), o3[0], o3[1], o3[2], int(q_ur[i]), int(q_r[i]), sc["inv_c1"])
    for k, name in enumerate(("slope_u", "ubar", "var_u", "lin_sse")):
        _same_bits(outs[k], host[:, k], "pgb_lin_fit/" + name)


# ------------------------------------------------------------------------------------------------
# 16-bit order keys of a split column (pgb_set_data builds them for matrices beyond the Infinity Cache; k_rows<F32>)
@pytest.mark.parametrize("kind", ["normal", "heavy_ties", "missing_and_inf", "constant", "tiny", "few_distinct_wide_range"])
def test_order_keys_decide_like_the_values_or_abstain(hip, kind):
    """key(x) < key(v) must imply x < v, key(x) > key(v) must imply x > v (equal keys abstain: the row pass then
    compares the float64 values), a missing value has key 0xFFFF and nothing else has; on a continuous column the
    bins are equi-depth, i.e. a value shares its key with about n / 65 535 rows."""
    rng = np.random.default_rng(7)
    n = 200_000
    if kind == "normal":
        x = rng.standard_normal(n)
    elif kind == "heavy_ties":
        x = rng.integers(0, 7, n).astype(float)
    elif kind == "missing_and_inf":
        x = rng.standard_normal(n) * 1e30
        x[rng.random(n) < 0.1] = np.nan
        x[:5] = [np.inf, -np.inf, 0.0, -0.0, 1e308]
    elif kind == "constant":
        x = np.full(n, 3.25)
    elif kind == "tiny":
        x = np.array([2.0, np.nan, -1.0])
        n = 3
    else:
        x = np.where(rng.random(n) < 0.5, 1e-300, 1.0) * rng.integers(1, 4, n)
    x = np.ascontiguousarray(x, float)
    keys = np.zeros(n, np.uint16)
    f = _fn(hip.lib.lib, "pgbh_order_keys", C.c_void_p, C.c_int64, C.c_void_p)
    assert f(x.ctypes.data, n, keys.ctypes.data) == 0
    miss = np.isnan(x)
    assert np.array_equal(keys == 0xFFFF, miss)
    xs, ks = x[~miss], keys[~miss].astype(np.int64)
    order = np.argsort(xs, kind="stable")
    assert np.all(np.diff(ks[order]) >= 0)                      # non-decreasing in x ...
    # ... which is the property the row pass uses: for any v, key(x) < key(v) => x < v and key(x) > key(v) => x > v
    for v_i in rng.integers(0, xs.size, 25):
        v, kv = xs[v_i], ks[v_i]
        assert np.all(xs[ks < kv] < v) and np.all(xs[ks > kv] > v)
    if kind == "normal":
        share = np.bincount(ks).max() / xs.size
        assert share < 20 / 65535                               # equi-depth: ~3 rows per key here, never dozens
