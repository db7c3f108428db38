"""Oracle-INDEPENDENT checks of the numeric layer both backends compile from include/pgbart_spec.h.

HIP == oracle is tautological for these functions (the same source is compiled into both sides), so
they are pinned here against SciPy / NumPy instead: the per-row log-likelihood
of every family (as DIFFERENCES in the linear predictor: each family drops its mu-free terms, which
cancel in the particle weights), the linear-leaf fit and its closed-form SSE, the leaf algebra, the
split-variable sampler and the subset split rule.
"""
import ctypes as C

import numpy as np
import pytest
from scipy import special, stats

from pymc_bart_amd import _abi


def _f(oracle, name, restype, *argtypes):
    fn = getattr(oracle.lib.lib, name)
    fn.restype, fn.argtypes = restype, list(argtypes)
    return fn


def _loglikq(oracle, family, y, mu, param=0.0, param2=1.0):
    f = _f(oracle, "pgbo_loglikq", None, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_double, C.c_double, C.c_void_p)
    y, mu = np.ascontiguousarray(y, float), np.ascontiguousarray(mu, float)
    out = np.zeros_like(y)
    f(_abi.FAMILIES[family], y.ctypes.data, mu.ctypes.data, y.size, param, param2, out.ctypes.data)
    return out


def _diff_check(oracle, family, y, ref_logpdf, param=0.0, param2=1.0, lo=-3.0, hi=3.0, tol=1e-10, seed=0):
    """ll(y; mu_a) - ll(y; mu_b) must equal the reference log-density difference (mu-free terms cancel)."""
    rng = np.random.default_rng(seed)
    mu_a, mu_b = rng.uniform(lo, hi, y.size), rng.uniform(lo, hi, y.size)
    la, lb = _loglikq(oracle, family, y, mu_a, param, param2), _loglikq(oracle, family, y, mu_b, param, param2)
    assert np.all(la <= 0.0) and np.all(la >= -2047.0)        # the contract's range (it clamps there)
    ok = (la > -2047.0) & (lb > -2047.0)
    assert ok.mean() > 0.9
    got = (la - lb)[ok]
    want = (ref_logpdf(y, mu_a) - ref_logpdf(y, mu_b))[ok]
    assert np.max(np.abs(got - want) / (1.0 + np.abs(want))) < tol, family


def test_bernoulli_links_against_scipy(oracle):
    rng = np.random.default_rng(1)
    y = (rng.random(4000) < 0.4).astype(float)
    mu = rng.uniform(-8, 8, 4000)
    s = np.where(y > 0.5, mu, -mu)
    assert np.max(np.abs(_loglikq(oracle, "bernoulli_probit", y, mu) - special.log_ndtr(s))) < 1e-12
    assert np.max(np.abs(_loglikq(oracle, "bernoulli_logit", y, mu) + np.logaddexp(0.0, -s))) < 1e-12


def test_asymmetric_laplace_is_the_quantile_regression_density(oracle):
    """-rho_q((y - mu)/b) is the Yu-Moyeed ALD: scipy's laplace_asymmetric with kappa^2 = q/(1-q),
    scale = b / sqrt(q (1 - q))."""
    y = np.random.default_rng(2).normal(0, 2, 3000)
    for b, q in ((0.25, 0.9), (1.0, 0.5), (2.0, 0.1)):
        kappa, scale = np.sqrt(q / (1 - q)), b / np.sqrt(q * (1 - q))
        _diff_check(oracle, "asymmetric_laplace", y,
                    lambda yy, m, k=kappa, sc=scale: stats.laplace_asymmetric.logpdf(yy, k, loc=m, scale=sc),
                    param=b, param2=q)


def test_student_t_against_scipy(oracle):
    y = np.random.default_rng(3).standard_t(3, 3000)
    for sigma, nu in ((0.2, 3.0), (1.0, 4.0), (2.5, 30.0)):
        _diff_check(oracle, "student_t", y, lambda yy, m, s=sigma, v=nu: stats.t.logpdf(yy, v, loc=m, scale=s),
                    param=sigma, param2=nu)


def test_gamma_log_link_against_scipy(oracle):
    """y ~ Gamma(shape alpha, mean exp(mu))."""
    y = np.random.default_rng(4).gamma(2.0, 1.5, 3000)
    for alpha in (0.7, 3.0, 12.0):
        _diff_check(oracle, "gamma_log", y,
                    lambda yy, m, a=alpha: stats.gamma.logpdf(yy, a, scale=np.exp(m) / a), param=alpha, tol=1e-9)


def test_count_families_against_scipy(oracle):
    rng = np.random.default_rng(5)
    y = rng.poisson(4.0, 3000).astype(float)
    _diff_check(oracle, "poisson_log", y, lambda yy, m: stats.poisson.logpmf(yy, np.exp(m)), lo=-1.0, tol=1e-9)
    for alpha in (0.5, 2.0, 9.0):  # NB2: mean exp(mu), dispersion alpha (PyMC's mu / alpha parameterisation)
        _diff_check(oracle, "negbin_log", y,
                    lambda yy, m, a=alpha: stats.nbinom.logpmf(yy, a, a / (a + np.exp(m))), param=alpha, lo=-1.0, tol=1e-9)


def test_multi_output_families_against_scipy(oracle):
    f = _f(oracle, "pgbo_loglik_multi", None, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
    rng = np.random.default_rng(6)
    n = 2000
    for K in (2, 3, 4, 6, 8):
        mu = np.ascontiguousarray(rng.normal(0, 2, (n, K)))
        y = rng.integers(0, K, n).astype(float)
        out = np.zeros(n)
        f(_abi.FAMILIES["categorical"], K, y.ctypes.data, mu.ctypes.data, n, out.ctypes.data)
        assert np.max(np.abs(out - special.log_softmax(mu, axis=1)[np.arange(n), y.astype(int)])) < 1e-12
    mu = np.ascontiguousarray(np.stack([rng.normal(0, 1, n), rng.normal(0, 1.5, n)], axis=1))
    y = rng.normal(0, 1, n)
    out = np.zeros(n)
    f(_abi.FAMILIES["normal_meanscale"], 2, y.ctypes.data, mu.ctypes.data, n, out.ctypes.data)
    want = stats.norm.logpdf(y, mu[:, 0], np.abs(mu[:, 1])) + 0.5 * np.log(2 * np.pi)
    ok = want > -2000  # the contract clamps at -2047
    assert np.max(np.abs(out[ok] - want[ok])) < 1e-10


def test_factorised_softmax_of_constant_leaves_against_scipy(oracle):
    """``pgb_loglik_cat_f`` (round 5): the row part (E_k, a_c) times the (particle, child) part (w_k, d_k), one
    logarithm per evaluation; the unfactorised form below S = 2^-200.  Against ``scipy.special.log_softmax`` of
    mu = eta + v over ordinary and over extreme spreads (the fallback regime), for K = 2..16: the same absolute
    accuracy as the unfactorised routine, plus the rounding of eta + v itself at large magnitudes."""
    f = _f(oracle, "pgbo_loglik_cat_f", None, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
    rng = np.random.default_rng(11)
    n = 20000
    for K in (2, 3, 4, 7, 16):
        for scale_eta, scale_v in ((0.3, 0.1), (3.0, 1.0), (30.0, 5.0), (300.0, 0.5), (1.0, 400.0), (500.0, 500.0)):
            eta = np.ascontiguousarray(rng.normal(0, scale_eta, (n, K)))
            v = rng.normal(0, scale_v, K)
            y = rng.integers(0, K, n).astype(float)
            out = np.zeros(n)
            f(K, y.ctypes.data, eta.ctypes.data, v.ctypes.data, n, out.ctypes.data)
            mu = eta + v
            want = np.clip(special.log_softmax(mu, axis=1)[np.arange(n), y.astype(int)], -2047.0, 0.0)
            tol = 2e-15 + 4 * np.spacing(np.abs(mu).max())  # (eta + v is rounded once before either form sees it)
            assert np.max(np.abs(out - want)) < tol * max(1.0, K / 4), (K, scale_eta, scale_v)
            assert out.max() <= 0.0 and out.min() >= -2047.0
    # the two regimes meet: rows whose factorised sum is just above / below the threshold give the same values as
    # the unfactorised routine to the last few ulp of log S
    g = _f(oracle, "pgbo_loglik_multi", None, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p)
    K = 3
    eta = np.zeros((4000, K))
    eta[:, 1] = -np.linspace(120.0, 160.0, 4000)     # a_1 sweeps across -138.6 = log 2^-200 while the child's d
    eta[:, 2] = -900.0                               # puts everything else far below: S ~ exp(a_1)
    v = np.array([-800.0, 0.0, -1000.0])
    y = np.ones(4000)
    a, b = np.zeros(4000), np.zeros(4000)
    f(K, y.ctypes.data, eta.ctypes.data, v.ctypes.data, 4000, a.ctypes.data)
    mu = np.ascontiguousarray(eta + v)
    g(_abi.FAMILIES["categorical"], K, y.ctypes.data, mu.ctypes.data, 4000, b.ctypes.data)
    assert np.max(np.abs(a - b)) < 1e-12 and np.all(np.abs(a) < 1e-9)   # class 1 holds all the mass: ll ~ 0


# ---------------------------------------------------------------------------------------------
def _scales(oracle, n, range_exp):
    f = _f(oracle, "pgbo_scales", None, C.c_int64, C.c_int, C.c_void_p)
    o = np.zeros(6)
    f(n, range_exp, o.ctypes.data)
    return dict(c1=o[0], c2=o[1], inv_c1=o[3], inv_c2=o[4])


def _q(x, scale):  # round-half-even fixed point, as pgb_quant inside its range
    return np.rint(np.asarray(x, float) * scale).astype(np.int64)


def test_leaf_algebra_against_numpy(oracle):
    leaf_sse = _f(oracle, "pgbo_leaf_sse", C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double)
    leaf_val = _f(oracle, "pgbo_leaf_value", C.c_double, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double, C.c_double)
    rng = np.random.default_rng(7)
    sc = _scales(oracle, 100_000, 6)
    for cnt in (1, 2, 57, 4000):
        r = rng.normal(0, 3, cnt)
        st = rng.normal(5, 2, cnt)
        v = float(rng.normal())
        sse = leaf_sse(cnt, int(_q(r, sc["c1"]).sum()), int(_q(r * r, sc["c2"]).sum()), v, sc["inv_c1"], sc["inv_c2"])
        assert sse == pytest.approx(float(np.sum((r - v) ** 2)), rel=1e-9, abs=1e-6)
        z, sd, m = 0.3, 0.8, 50.0
        got = leaf_val(cnt, int(_q(st, sc["c1"]).sum()), sc["inv_c1"], m, z, sd)
        assert got == pytest.approx(st.mean() / m + z * sd, rel=1e-10, abs=1e-9)
    assert leaf_val(0, 0, sc["inv_c1"], 50.0, 0.3, 0.8) == 0.0   # [U] empty child -> 0


def test_linear_leaf_fit_and_its_sse_against_polyfit(oracle):
    """[U] fast_linear_fit: slope of sum_trees / m on x over the leaf's rows; the leaf predicts
    value + slope (x - xbar).  pgb_lin_fit works on u = x 2^-ex and fixed-point sums; pgb_lin_sse gives the
    SSE of that prediction in closed form.  Both against np.polyfit / a direct sum."""
    lin_fit = _f(oracle, "pgbo_lin_fit", None, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_int64, C.c_double,
                 C.c_double, C.c_double, C.c_void_p)
    lin_sse = _f(oracle, "pgbo_lin_sse", C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int64,
                 C.c_int64, C.c_double)
    leaf_sse = _f(oracle, "pgbo_leaf_sse", C.c_double, C.c_int64, C.c_int64, C.c_int64, C.c_double, C.c_double, C.c_double)
    col_ex = _f(oracle, "pgbo_col_exponent", C.c_int, C.c_double)
    rng = np.random.default_rng(8)
    range_exp, m = 6, 20.0
    sc = _scales(oracle, 50_000, range_exp)
    R, inv_R = 2.0 ** (range_exp - 1), 2.0 ** (1 - range_exp)
    for cnt, xscale in ((3, 1.0), (40, 0.01), (900, 37.0), (5000, 1.0)):
        x = rng.normal(0.3, 1.0, cnt) * xscale
        st = 4.0 + 1.7 * x / xscale + rng.normal(0, 0.5, cnt)       # sum_trees on the leaf's rows
        r = rng.normal(0, 1, cnt) + 0.9 * x / xscale                  # residuals y - sum_trees_noi
        ex = col_ex(float(np.abs(x).max()))
        assert np.abs(x).max() <= 2.0 ** ex
        u = x * 2.0 ** -ex
        qs = [int(_q(u * R, sc["c1"]).sum()), int(_q(u * u * R, sc["c1"]).sum()), int(_q(u * st, sc["c1"]).sum()),
              int(_q(st, sc["c1"]).sum())]
        out = np.zeros(3)
        lin_fit(cnt, qs[0], qs[1], qs[2], qs[3], sc["inv_c1"], inv_R, m, out.ctypes.data)
        slope_x = out[0] * 2.0 ** -ex                                  # what the leaf stores: d value / d x
        want = np.polyfit(x, st / m, 1)[0]
        assert slope_x == pytest.approx(want, rel=2e-6, abs=1e-9), cnt
        assert out[1] * 2.0 ** ex == pytest.approx(x.mean(), rel=1e-7, abs=1e-9)
        # SSE of r around value + slope (x - xbar), from the constant-leaf SSE
        value = float(rng.normal())
        sse_c = leaf_sse(cnt, int(_q(r, sc["c1"]).sum()), int(_q(r * r, sc["c2"]).sum()), value, sc["inv_c1"], sc["inv_c2"])
        sse_l = lin_sse(sse_c, out[0], out[1], out[2], int(_q(u * r, sc["c1"]).sum()), int(_q(r, sc["c1"]).sum()),
                        sc["inv_c1"])
        direct = float(np.sum((r - value - slope_x * (x - x.mean())) ** 2))
        assert sse_l == pytest.approx(direct, rel=1e-6, abs=1e-5), cnt
    out = np.zeros(3)
    lin_fit(2, 1, 1, 1, 1, sc["inv_c1"], inv_R, m, out.ctypes.data)       # [U] fewer than 3 rows: constant leaf
    assert out[0] == 0.0
    x = np.full(50, 0.25)                                                  # no spread: constant leaf
    lin_fit(50, int(_q(x * R, sc["c1"]).sum()), int(_q(x * x * R, sc["c1"]).sum()), 7, 9, sc["inv_c1"], inv_R, m,
            out.ctypes.data)
    assert out[0] == 0.0


def test_split_variable_sampler_is_the_inverse_cdf_of_the_weights(oracle):
    """[U] SampleSplittingVariable.rvs: P(j) proportional to the (integer) split weights."""
    sample = _f(oracle, "pgbo_sample_var", C.c_int, C.c_void_p, C.c_int, C.c_double)
    a_init = _f(oracle, "pgbo_alpha_init", C.c_int64, C.c_double, C.c_double)
    a_unit = _f(oracle, "pgbo_alpha_unit", C.c_int64, C.c_double)
    prior = np.array([3.0, 1.0, 1.0, 0.5, 2.0, 0.25])
    A = np.array([a_init(float(v), float(prior.max())) for v in prior], np.int64)
    assert np.allclose(A / A.sum(), prior / prior.sum(), rtol=1e-6)
    unit = a_unit(float(prior.max()))
    assert unit / A[1] == pytest.approx(1.0, rel=1e-6)      # one tuning count == one unit of prior weight
    S = np.cumsum(A)
    us = np.random.default_rng(9).random(20000)
    got = np.array([sample(S.ctypes.data, len(S), float(u)) for u in us])
    want = np.searchsorted(S.astype(float), us * float(S[-1]), side="left")
    assert np.array_equal(got, np.minimum(want, len(S) - 1))
    freq = np.bincount(got, minlength=len(S)) / len(us)
    assert np.max(np.abs(freq - prior / prior.sum())) < 0.01
    assert sample(S.ctypes.data, len(S), 0.0) == 0 and sample(S.ctypes.data, len(S), 1.0 - 2 ** -53) == len(S) - 1


def test_subset_rule_masks_and_membership(oracle):
    """SubsetSplitRule (reference bart.py:100-103; deviation 10): the chosen row's category always goes left,
    every other category with probability 1/2, membership is a bit test; continuous / one-hot rules."""
    subset_value = _f(oracle, "pgbo_subset_value", C.c_double, C.c_double, C.c_double)
    go_left = _f(oracle, "pgbo_go_left", C.c_int, C.c_int, C.c_double, C.c_double)
    rng = np.random.default_rng(10)
    counts = np.zeros(8)
    N = 4000
    for _ in range(N):
        cat = int(rng.integers(0, 8))
        v = subset_value(float(rng.random()), float(cat))
        mask = int(v)
        assert float(mask) == v and 0 < mask < 2 ** 52                       # exact integer bit mask
        assert (mask >> cat) & 1 == 1 and go_left(_abi.RULE_SUBSET, float(cat), v) == 1
        for other in range(8):
            member = go_left(_abi.RULE_SUBSET, float(other), v)
            assert member == ((mask >> other) & 1)
            if other != cat:
                counts[other] += member
    assert np.all(np.abs(counts / (N * 7 / 8) - 0.5) < 0.04)                   # fair coins for the others
    assert go_left(_abi.RULE_SUBSET, 51.0, subset_value(0.0, 51.0)) == 1      # last representable category
    assert go_left(_abi.RULE_CONTINUOUS, 1.0, 1.0) == 1 and go_left(_abi.RULE_CONTINUOUS, 1.0000001, 1.0) == 0
    assert go_left(_abi.RULE_ONEHOT, 2.0, 2.0) == 1 and go_left(_abi.RULE_ONEHOT, 2.0, 3.0) == 0
