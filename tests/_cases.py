"""Seeded parity cases shared by the golden-fixture generator and the GPU parity tests."""
from __future__ import annotations

import hashlib

import numpy as np

from pymc_bart_amd.sampler import PyBartSettings, PySampler


def make_case(name: str):
    if name.startswith("upstream/"):  # the same case under the upstream-semantics switches (PGB_COMPAT_*: both bits)
        c = make_case(name[len("upstream/"):])
        c.update(name=name, compat=3)
        return c
    rng = np.random.default_rng(abs(hash(name)) % (2 ** 31) if False else int(hashlib.sha1(name.encode()).hexdigest()[:8], 16))
    c = dict(name=name, m=10, P=10, steps=24, batch=(0.1, 0.1), rules=None, prior=None, seed=3415)
    if name == "cfg1_friedman":  # BASELINE.json configs[0]
        n, p = 500, 5
        X = rng.uniform(0, 1, (n, p))
        Y = (10 * np.sin(np.pi * X[:, 0] * X[:, 1]) + 20 * (X[:, 2] - 0.5) ** 2 + 10 * X[:, 3]
             + 5 * X[:, 4] + rng.normal(0, 1, n))
        c.update(m=50, P=10, steps=40)
    elif name == "nan_onehot_prior":
        n, p = 5000, 8
        X = rng.normal(size=(n, p))
        X[:, 6] = rng.integers(0, 5, n)
        X[:, 7] = rng.integers(0, 2, n)
        X[rng.random(n) < 0.1, 1] = np.nan
        X[rng.random(n) < 0.3, 6] = np.nan
        Y = X[:, 0] * 2 + np.where(X[:, 7] > 0, 1.5, -1.5) + rng.normal(0, 0.5, n)
        rules = np.zeros(p, np.int32)
        rules[6:] = 1
        c.update(m=20, P=20, steps=40, rules=rules, prior=np.array([3, 1, 1, 1, 1, 0.5, 2, 2.0]))
    elif name == "ragged_1025":
        n, p = 1025, 3
        X = rng.normal(size=(n, p))
        Y = np.abs(X[:, 0]) + rng.normal(0, 0.1, n)
        c.update(m=7, P=6, steps=30, batch=(3, 2))
    elif name == "tiny_n3":
        n, p = 3, 2
        X = rng.normal(size=(n, p))
        Y = np.array([0.0, 1.0, 5.0])
        c.update(m=4, P=4, steps=16)
    elif name == "one_tree_two_particles":
        n, p = 300, 2
        X = rng.normal(size=(n, p))
        Y = X[:, 0] + rng.normal(0, 0.1, n)
        c.update(m=1, P=2, steps=30)
    elif name == "max_particles":
        n, p = 2100, 4
        X = rng.normal(size=(n, p))
        Y = np.sin(3 * X[:, 0]) * 4 + rng.normal(0, 0.3, n)
        c.update(m=6, P=64, steps=12)
    elif name == "particles_128":  # beyond one particle per lane: the 128-particle build of the library (two per lane)
        n, p = 2600, 5
        X = rng.normal(size=(n, p))
        X[rng.random(n) < 0.08, 2] = np.nan
        Y = np.sin(3 * X[:, 0]) * 3 + X[:, 1] * np.nan_to_num(X[:, 2]) + rng.normal(0, 0.3, n)
        c.update(m=5, P=128, steps=12)
    elif name == "particles_100_probit":  # an odd count above 64, a per-row family
        n, p = 1800, 4
        X = rng.normal(size=(n, p))
        Y = (rng.random(n) < 1 / (1 + np.exp(-2 * X[:, 0] + X[:, 1]))).astype(float)
        c.update(m=4, P=100, steps=10, family="bernoulli_probit")
    elif name == "categorical_k4_particles_100":  # K-vector leaves above 64 particles: both blocks of 64 have jobs whose
        n, p, K = 3000, 6, 4                      # extension outputs the likelihood pass lists (one lane per pair) and hands on
        X = rng.normal(size=(n, p))
        F = np.stack([X[:, 0], -X[:, 0], 1.5 * X[:, 1], 0 * X[:, 0]])
        pr = np.exp(F) / np.exp(F).sum(0)
        Y = (rng.random(n)[None, :] > np.cumsum(pr, axis=0)).sum(0).clip(0, K - 1).astype(float)
        c.update(m=4, P=100, steps=10, family="categorical", K=K)
    elif name == "duplicates":
        n, p = 4096, 3
        X = rng.integers(0, 3, (n, p)).astype(float)  # heavy ties, no jitter at this level
        Y = X[:, 0] - X[:, 1] + rng.normal(0, 0.2, n)
        rules = np.array([0, 1, 1], np.int32)
        c.update(m=8, P=12, steps=30, rules=rules)
    elif name == "deep_trees":
        n, p = 3000, 3
        X = rng.normal(size=(n, p))
        Y = np.sin(5 * X[:, 0]) + np.cos(3 * X[:, 1]) + rng.normal(0, 0.05, n)
        c.update(m=5, P=10, steps=20, alpha=0.999, beta=0.3)
    elif name == "probit_cfg4_small":  # BASELINE.json configs[3] at test size
        from scipy.special import ndtr
        n, p = 6000, 10
        X = rng.normal(size=(n, p))
        f = 1.5 * X[:, 0] - (X[:, 1] > 0) + 0.5 * X[:, 2] * X[:, 0]
        Y = (rng.random(n) < ndtr(f / 1.0)).astype(float)
        c.update(m=20, P=12, steps=30, family="bernoulli_probit")
    elif name == "logit_nan_onehot":
        n, p = 3000, 5
        X = rng.normal(size=(n, p))
        X[:, 3] = rng.integers(0, 3, n)
        X[:, 4] = 1.0  # constant one-hot column: every split attempt on it fails
        X[rng.random(n) < 0.15, 0] = np.nan
        X[rng.random(n) < 0.2, 4] = np.nan
        f = np.where(np.isnan(X[:, 0]), 0.0, 2 * X[:, 0]) + (X[:, 3] == 1) * 1.5
        Y = (rng.random(n) < 1 / (1 + np.exp(-f))).astype(float)
        c.update(m=10, P=16, steps=30, family="bernoulli_logit", rules=np.array([0, 0, 0, 1, 1], np.int32))
    elif name == "categorical_k3_reference":  # reference tests/test_bart.py:140-164 (shape=(3, 9))
        Y = np.array([0, 0, 0, 1, 1, 1, 2, 2, 2], float)
        X = np.concatenate([Y[:, None], rng.integers(0, 6, size=(9, 4))], axis=1).astype(float)
        c.update(m=2, P=10, steps=60, family="categorical", K=3, rules=np.array([1] * 5, np.int32))
        n = 9
    elif name == "categorical_k4_cfg5_small":  # BASELINE.json configs[4] at test size
        n, p, K = 5000, 12, 4
        X = rng.normal(size=(n, p))
        X[rng.random(n) < 0.1, 2] = np.nan
        F = np.stack([X[:, 0], -X[:, 0], 1.5 * X[:, 1], 0 * X[:, 0]])
        pr = np.exp(F) / np.exp(F).sum(0)
        Y = (rng.random(n)[None, :] > np.cumsum(pr, axis=0)).sum(0).clip(0, K - 1).astype(float)
        c.update(m=15, P=12, steps=24, family="categorical", K=K)
    elif name == "categorical_k6_generic":  # more than 4 outputs: the run-time-K kernel instances
        n, p, K = 3000, 6, 6
        X = rng.normal(size=(n, p))
        X[rng.random(n) < 0.1, 1] = np.nan
        F = np.stack([X[:, 0], -X[:, 0], X[:, 2], -X[:, 2], 0.5 * X[:, 3], 0 * X[:, 0]])
        pr = np.exp(F) / np.exp(F).sum(0)
        Y = (rng.random(n)[None, :] > np.cumsum(pr, axis=0)).sum(0).clip(0, K - 1).astype(float)
        c.update(m=8, P=10, steps=20, family="categorical", K=K)
    elif name == "categorical_k12":  # beyond the former cap of 8 outputs (a 10-class softmax model raised): three tiles
        n, p, K = 2500, 7, 12
        X = rng.normal(size=(n, p))
        X[rng.random(n) < 0.1, 1] = np.nan
        X[:, 6] = rng.integers(0, 4, n)
        F = np.stack([np.cos(0.5 * k) * X[:, k % 4] + 0.3 * np.sin(k) * X[:, (k + 1) % 5] + 0.4 * (X[:, 6] == k % 4)
                      for k in range(K)])
        pr = np.exp(F) / np.exp(F).sum(0)
        Y = (rng.random(n)[None, :] > np.cumsum(pr, axis=0)).sum(0).clip(0, K - 1).astype(float)
        c.update(m=6, P=10, steps=16, family="categorical", K=K, rules=np.array([0, 0, 0, 0, 0, 0, 1], np.int32))
    elif name == "categorical_k16_linear":  # the largest K, linear leaves (run-time-K instances with LIN)
        n, p, K = 1500, 4, 16
        X = rng.normal(size=(n, p))
        F = np.stack([np.cos(0.4 * k) * X[:, k % 3] + 0.2 * k / K * X[:, 3] for k in range(K)])
        pr = np.exp(F) / np.exp(F).sum(0)
        Y = (rng.random(n)[None, :] > np.cumsum(pr, axis=0)).sum(0).clip(0, K - 1).astype(float)
        c.update(m=4, P=8, steps=12, family="categorical", K=K, response="linear")
    elif name in ("poisson_counts", "negbin_counts"):  # the count models of the PyMC-BART docs (log link)
        n, p = 4000, 5
        X = rng.normal(size=(n, p))
        X[rng.random(n) < 0.1, 2] = np.nan
        rate = np.exp(0.8 * X[:, 0] - 0.5 * (X[:, 1] > 0) + 1.0)
        if name == "poisson_counts":
            Y = rng.poisson(rate).astype(float)
            c.update(family="poisson_log")
        else:
            Y = rng.negative_binomial(2.0, 2.0 / (2.0 + rate)).astype(float)
            c.update(family="negbin_log", lik_params=[2.0])
        c.update(m=12, P=12, steps=30, bart_Y=np.log(Y + 0.5))
    elif name in ("linear_poisson", "mix_probit"):  # linear leaves with per-row families
        n, p = 3000, 3
        X = rng.uniform(-2, 2, size=(n, p))
        X[rng.random(n) < 0.1, 2] = np.nan
        f = np.where(X[:, 0] < 0, 1.0 * X[:, 0] + 0.5, -0.8 * X[:, 0] + 0.5)
        if name == "linear_poisson":
            Y = rng.poisson(np.exp(f)).astype(float)
            c.update(family="poisson_log", response="linear", bart_Y=np.log(Y + 0.5))
        else:
            from scipy.special import ndtr
            Y = (rng.random(n) < ndtr(f)).astype(float)
            c.update(family="bernoulli_probit", response="mix")
        c.update(m=8, P=12, steps=30)
    elif name == "gamma_positive":  # positive continuous response, log link
        n, p = 3000, 4
        X = rng.normal(size=(n, p))
        mean = np.exp(0.6 * X[:, 0] - 0.4 * (X[:, 1] > 0) + 0.5)
        Y = rng.gamma(3.0, mean / 3.0)
        c.update(family="gamma_log", lik_params=[3.0], m=10, P=12, steps=30, bart_Y=np.log(Y))
    elif name == "poisson_exposure":  # per-row offset of the linear predictor (log-exposure of a count model)
        n, p = 3000, 4
        X = rng.normal(size=(n, p))
        expo = rng.uniform(0.2, 5.0, n)
        lograte = 0.7 * X[:, 0] + 0.5
        Y = rng.poisson(expo * np.exp(lograte)).astype(float)
        c.update(family="poisson_log", m=10, P=12, steps=30, bart_Y=np.log((Y + 0.5) / expo), offset=np.log(expo))
    elif name in ("quantile_asymlaplace", "robust_student_t"):  # two-parameter per-row families
        n, p = 3500, 4
        X = rng.uniform(-2, 2, size=(n, p))
        X[rng.random(n) < 0.1, 3] = np.nan
        f = np.sin(2 * X[:, 0]) + 0.5 * X[:, 1]
        if name == "quantile_asymlaplace":
            Y = f + rng.normal(0, 0.2 + 0.3 * (X[:, 0] > 0), n)      # heteroscedastic: quantiles differ from the mean
            c.update(family="asymmetric_laplace", lik_params=[0.25, 0.9])
        else:
            Y = f + 0.2 * rng.standard_t(3, n)                        # heavy tails
            c.update(family="student_t", lik_params=[0.2, 3.0])
        c.update(m=10, P=12, steps=30)
    elif name in ("linear_response", "mix_response"):  # reference tests parametrise response=["constant","linear"]
        n, p = 3000, 4
        X = rng.uniform(-2, 2, size=(n, p))
        X[:, 3] = np.round(X[:, 3])            # ties
        X[rng.random(n) < 0.1, 1] = np.nan     # missing values in a regressor
        f = np.where(X[:, 0] < 0, 2 * X[:, 0] + 1, -1.5 * X[:, 0] + 1) + 0.5 * np.nan_to_num(X[:, 1])
        Y = f + rng.normal(0, 0.1, n)
        c.update(m=8, P=12, steps=30, response="linear" if name == "linear_response" else "mix")
    elif name in ("linear_mixed_rules", "mix_probit_mixed_rules", "categorical_k3_linear_mixed_rules"):
        # linear / mix leaves next to OneHot / Subset columns: a leaf regresses on whatever column its parent
        # split on (upstream: fast_linear_fit on X[idx, selected_predictor], whatever the rule, bart.py:88-103)
        n, p = 3000, 5
        X = rng.uniform(-2, 2, size=(n, p))
        X[:, 2] = rng.integers(0, 3, n)        # one-hot column
        X[:, 3] = rng.integers(0, 9, n)        # subset column, 9 categories
        X[:, 4] = 1.0                          # subset column with a single category (every split fails)
        X[rng.random(n) < 0.1, 1] = np.nan
        X[rng.random(n) < 0.1, 3] = np.nan
        f = np.where(X[:, 0] < 0, 2 * X[:, 0] + 1, -1.5 * X[:, 0] + 1) + (X[:, 2] == 1) + 0.3 * np.nan_to_num(X[:, 3])
        c.update(m=8, P=12, steps=30, rules=np.array([0, 0, 1, 2, 2], np.int32), prior=np.array([1.0, 1.0, 2.0, 2.0, 1.0]))
        if name == "linear_mixed_rules":
            Y = f + rng.normal(0, 0.2, n)
            c.update(response="linear")
        elif name == "mix_probit_mixed_rules":
            from scipy.special import ndtr
            Y = (rng.random(n) < ndtr(f - 1.0)).astype(float)
            c.update(family="bernoulli_probit", response="mix")
        else:
            K = 3
            F = np.stack([f, -f, 0.5 * (X[:, 2] == 2)])
            pr = np.exp(F) / np.exp(F).sum(0)
            Y = (rng.random(n)[None, :] > np.cumsum(pr, axis=0)).sum(0).clip(0, K - 1).astype(float)
            c.update(family="categorical", K=K, response="linear", steps=20)
    elif name.startswith("stump_first_"):
        # deviation 12 on the per-row families: tiny forest, three particles, and a key under which the untouched
        # stump wins the first tree updates -- the running sd of the accepted predictions is exactly 0 when leaf_sd
        # is first tuned, in k_ctrl AND in the likelihood pass's own copy of that rule (a 429-of-11 674 fuzz
        # divergence in round 3 when only one of the two had it)
        r77 = np.random.default_rng(77)
        n, p = 600, 3
        X = r77.normal(size=(n, p))
        f = 1.2 * X[:, 0]
        Yb = (r77.random(n) < 1 / (1 + np.exp(-f))).astype(float)
        Yc = np.clip(np.round(f + r77.normal(0, 0.5, n) + 1), 0, 2)
        Yp = r77.poisson(np.exp(f)).astype(float)
        c.update(m=2, P=3, steps=14, batch=(0.5, 0.5))
        if name == "stump_first_probit":
            Y = Yb
            c.update(family="bernoulli_probit", seed=0)
        elif name == "stump_first_categorical":
            Y = Yc
            c.update(family="categorical", K=3, seed=3)
        else:
            Y = Yp
            c.update(family="poisson_log", seed=0, bart_Y=np.log(Yp + 0.5))
    elif name == "meanscale_k2_reference":  # reference tests/test_bart.py:107-123 (shape=(2, 250))
        n, p = 250, 3
        X = rng.normal(0, 1, size=(n, p))
        Y = rng.normal(0, 1, size=n) * (0.5 + (X[:, 0] > 0)) + X[:, 1]
        c.update(m=2, P=10, steps=60, family="normal_meanscale", K=2)
    elif name == "meanscale_k2_linear":  # reference tests/test_bart.py:107-123 with response="linear"
        n, p = 250, 3
        X = rng.normal(0, 1, size=(n, p))
        Y = rng.normal(0, 1, size=n) * (0.5 + (X[:, 0] > 0)) + X[:, 1]
        c.update(m=2, P=10, steps=60, family="normal_meanscale", K=2, response="linear")
    elif name == "categorical_k3_mix":  # K-vector leaves with a slope per output, "mix", missing values
        n, p, K = 3000, 4, 3
        X = rng.uniform(-2, 2, size=(n, p))
        X[rng.random(n) < 0.1, 1] = np.nan
        F = np.stack([1.2 * X[:, 0], -1.2 * X[:, 0], 0.5 * np.nan_to_num(X[:, 1])])
        pr = np.exp(F) / np.exp(F).sum(0)
        Y = (rng.random(n)[None, :] > np.cumsum(pr, axis=0)).sum(0).clip(0, K - 1).astype(float)
        c.update(m=8, P=12, steps=24, family="categorical", K=K, response="mix")
    elif name == "categorical_k3_offset":  # additive multi-output model: per-row, per-output offsets of the predictors
        n, p, K = 2500, 4, 3
        X = rng.normal(size=(n, p))
        Z = rng.normal(size=n)                                   # a covariate handled by another model term
        off = np.stack([0.8 * Z, -0.8 * Z, np.zeros(n)])
        F = np.stack([X[:, 0], -X[:, 0], 0.7 * X[:, 1]]) + off
        pr = np.exp(F) / np.exp(F).sum(0)
        Y = (rng.random(n)[None, :] > np.cumsum(pr, axis=0)).sum(0).clip(0, K - 1).astype(float)
        c.update(m=8, P=10, steps=24, family="categorical", K=K, offset=off)
    elif name == "onehot_fail_nan":  # failed one-hot splits that shed NaN rows (Normal family)
        n, p = 2500, 3
        X = rng.normal(size=(n, p))
        X[:, 1] = 2.0
        X[rng.random(n) < 0.25, 1] = np.nan
        X[:, 2] = rng.integers(0, 2, n)
        Y = X[:, 0] + (X[:, 2] > 0) + rng.normal(0, 0.3, n)
        c.update(m=6, P=10, steps=30, rules=np.array([0, 1, 1], np.int32), prior=np.array([1.0, 3.0, 1.0]))
    elif name == "subset_rule":  # SubsetSplitRule (bart.py:100-103): categorical codes, set-valued splits
        n, p = 4000, 5
        X = rng.normal(size=(n, p))
        X[:, 2] = rng.integers(0, 7, n)       # 7 categories, effect of the set {1, 4, 6}
        X[:, 3] = rng.integers(0, 2, n)       # binary category
        X[:, 4] = 3.0                         # a single category: every subset split on it fails
        X[rng.random(n) < 0.1, 2] = np.nan
        X[rng.random(n) < 0.2, 4] = np.nan
        Y = (np.isin(X[:, 2], [1, 4, 6]) * 3.0 + (X[:, 3] > 0) * 1.0 + 0.5 * X[:, 0]
             + rng.normal(0, 0.3, n))
        c.update(m=12, P=14, steps=30, rules=np.array([0, 0, 2, 2, 2], np.int32),
                 prior=np.array([1.0, 1.0, 3.0, 1.0, 1.0]))
    else:
        raise KeyError(name)
    c.update(X=X, Y=Y)
    return c


CASES = ["cfg1_friedman", "nan_onehot_prior", "ragged_1025", "tiny_n3", "one_tree_two_particles",
         "max_particles", "particles_128", "particles_100_probit", "categorical_k4_particles_100", "duplicates", "deep_trees", "onehot_fail_nan", "probit_cfg4_small",
         "logit_nan_onehot", "categorical_k3_reference", "categorical_k4_cfg5_small",
         "meanscale_k2_reference", "subset_rule", "categorical_k6_generic", "categorical_k12", "categorical_k16_linear", "linear_response", "mix_response", "poisson_counts", "negbin_counts", "quantile_asymlaplace", "robust_student_t", "poisson_exposure", "gamma_positive", "linear_poisson", "mix_probit",
         "meanscale_k2_linear", "categorical_k3_mix", "categorical_k3_offset",
         "linear_mixed_rules", "mix_probit_mixed_rules", "categorical_k3_linear_mixed_rules",
         "stump_first_probit", "stump_first_categorical", "stump_first_poisson",
         # pgb_settings.compat = 3: fresh particles at log-weight 0, empty right leaves of one-hot splits
         "upstream/cfg1_friedman", "upstream/nan_onehot_prior", "upstream/onehot_fail_nan", "upstream/probit_cfg4_small",
         "upstream/categorical_k3_reference", "upstream/subset_rule", "upstream/linear_mixed_rules",
         "upstream/categorical_k3_linear_mixed_rules", "upstream/particles_128", "upstream/logit_nan_onehot"]


def run_case(c, backend, record_every: int = 1, checkpoint_at=()):
    """Run the case on a backend; returns everything the two backends must agree on.
    ``checkpoint_at``: step indices before which the chain is checkpointed, its sampler destroyed
    and a freshly built sampler restored from the image (must not change anything); a dict
    ``{step: backend}`` moves the chain to ANOTHER backend there (the image belongs to none:
    include/pgbart_image.h)."""
    X, Y = c["X"], c["Y"]
    p = X.shape[1]
    family = c.get("family", "normal")
    st = PyBartSettings.from_data(X, c.get("bart_Y", Y), m=c["m"], num_particles=c["P"], seed=c["seed"], batch=c["batch"],
                                  alpha=c.get("alpha", 0.95), beta=c.get("beta", 2.0), family=family,
                                  n_outputs=c.get("K", 1), response=c.get("response", "constant"),
                                  compat=c.get("compat", 0))
    rules = np.zeros(p, np.int32) if c["rules"] is None else c["rules"]
    prior = np.ones(p) if c["prior"] is None else c["prior"]
    s = PySampler(st, X, Y, rules, prior, backend=backend)
    if c.get("offset") is not None:
        s.set_offset(c["offset"])
    sig_rng = np.random.default_rng(99)
    sums, vis, trees = [], [], []
    half = c["steps"] // 2
    for it in range(c["steps"]):
        if it in checkpoint_at:
            blob = s.checkpoint()
            del s
            if isinstance(checkpoint_at, dict):
                backend = checkpoint_at[it]
            s = PySampler(st, X, Y, rules, prior, backend=backend)
            if c.get("offset") is not None:
                s.set_offset(c["offset"])
            s.restore(blob)
        sig = float(0.5 + sig_rng.random())  # sigma moves like a Gibbs/NUTS neighbour
        s.set_likelihood([sig] if family == "normal" else c.get("lik_params", []))
        stv, vi = s.step(tune=it < half)
        if it % record_every == 0:
            sums.append(stv)
            vis.append(vi)
            ta = s.export_trees(0)
            parts = [ta.tree_id, ta.node_off, ta.var, ta.left, ta.right, ta.count, ta.split.view(np.int64),
                     ta.value.ravel().view(np.int64)]
            if c.get("response", "constant") != "constant":  # linear leaves are part of the fingerprint
                parts += [ta.slope.ravel().view(np.int64), ta.xbar.view(np.int64), ta.svar]
            trees.append(np.concatenate(parts))
    forest = s.export_trees(1)
    ctr = s.counters.as_dict()
    ctr.pop("slots")
    return dict(sum_trees=np.array(sums), vi=np.array(vis), trees=trees, forest=forest, counters=ctr,
                state=s.state(), split_weights=s.split_weights(), sampler=s)


def digest(res) -> dict:
    """Compact, exact fingerprint of a run (what the golden fixture stores)."""
    h = hashlib.sha256()
    h.update(res["sum_trees"].tobytes())
    h.update(res["vi"].astype(np.int32).tobytes())
    for t in res["trees"]:
        h.update(np.ascontiguousarray(t).tobytes())
    f = res["forest"]
    for a in (f.var, f.left, f.right, f.count, f.split, f.value):
        h.update(np.ascontiguousarray(a).tobytes())
    h.update(res["split_weights"].tobytes())
    h.update(np.asarray(res["state"]["leaf_sd"]).tobytes())
    return {
        "sha256": h.hexdigest(),
        "counters": {k: int(v) for k, v in res["counters"].items()},
        "last_sum_trees_head": np.asarray(res["sum_trees"][-1]).ravel()[:8].tolist(),
        "leaf_sd": float(res["state"]["leaf_sd"][0]),
        "iter": int(res["state"]["iter"]),
    }


def random_case(seed, large=False, compat=0):
    """A random configuration for the fuzz parity test: sizes around the chunk / wave boundaries,
    every family, every split rule, NaNs, ties, priors, batch sizes, alpha / beta.
    `large`: hundreds of chunks per pass (work items beyond one per workgroup, particle groups of
    more than one particle), few trees and steps so that the oracle still answers in seconds."""
    rng = np.random.default_rng(seed)
    fam = rng.choice(["normal", "normal", "normal", "bernoulli_probit", "bernoulli_logit", "categorical", "normal_meanscale",
                      "poisson_log", "negbin_log", "asymmetric_laplace", "student_t", "gamma_log"])
    n = int(rng.choice([3, 17, 255, 256, 257, 1023, 1024, 1025, 2049, 5000, 20000]))
    p = int(rng.integers(1, 9))
    m = int(rng.integers(1, 12))
    P = int(rng.choice([2, 3, 5, 10, 20, 40, 64, 65, 97, 128]))  # (> 64: the two-particles-per-lane build)
    if large:
        n = int(rng.choice([50_000, 131_072, 200_001, 400_000, 1_048_577]))
        m = int(rng.integers(1, 5))
        P = int(rng.choice([5, 20, 40, 64, 128]))
    X = rng.normal(size=(n, p))
    rules = np.zeros(p, np.int32)
    for j in range(p):
        r = rng.random()
        if r < 0.2:
            X[:, j] = rng.integers(0, int(rng.integers(1, 6)), n); rules[j] = 1
        elif r < 0.35:
            X[:, j] = rng.integers(0, int(rng.integers(1, 9)), n); rules[j] = 2
        elif r < 0.45:
            X[:, j] = np.round(X[:, j])  # heavy ties, continuous rule
        if rng.random() < 0.3:
            X[rng.random(n) < rng.uniform(0.01, 0.5), j] = np.nan
    f = np.nan_to_num(X[:, 0]) * 1.5 + (np.nan_to_num(X[:, -1]) > 0)
    K = 1
    if fam == "normal":
        Y = f + rng.normal(0, 0.5, n)
    elif fam.startswith("bernoulli"):
        Y = (rng.random(n) < 1 / (1 + np.exp(-f))).astype(float)
    elif fam in ("poisson_log", "negbin_log"):
        Y = rng.poisson(np.exp(np.clip(f, -3, 3))).astype(float)
    elif fam in ("asymmetric_laplace", "student_t"):
        Y = f + rng.standard_t(3, n) * 0.5
    elif fam == "gamma_log":
        Y = rng.gamma(2.0, np.exp(np.clip(f, -3, 3)) / 2.0) + 1e-6
    elif fam == "categorical":
        # classes that DEPEND on the covariates (Gumbel-max over logits that fan out with f): with a pure-noise
        # response the stump wins nearly every update and the K-vector growth paths are hardly exercised
        K = int(rng.integers(2, 17)) if rng.random() < 0.35 else int(rng.integers(2, 8))  # (up to PGB_MAX_OUTPUTS = 16)
        logits = np.stack([f * (k - 0.5 * (K - 1)) for k in range(K)]) + rng.gumbel(size=(K, n))
        Y = np.argmax(logits, axis=0).astype(float)
    else:
        K = 2; Y = f + rng.normal(0, 1, n) * (0.5 + (np.nan_to_num(X[:, 0]) > 0))
    batch = (float(rng.choice([0.1, 0.34, 1.0])), float(rng.choice([0.1, 0.5])))
    response = "constant"
    if not rules.any():
        response = str(rng.choice(["constant", "linear", "mix"]))
    elif rng.random() < 0.25:  # linear / mix leaves next to one-hot / subset columns
        response = str(rng.choice(["linear", "mix"]))
    extra = {}
    if fam in ("poisson_log", "negbin_log"):
        extra["bart_Y"] = np.log(Y + 0.5)
        if fam == "negbin_log":
            extra["lik_params"] = [float(rng.uniform(0.3, 5.0))]
    if fam == "gamma_log":
        extra["bart_Y"] = np.log(Y)
        extra["lik_params"] = [float(rng.uniform(0.5, 5.0))]
    if fam == "asymmetric_laplace":
        extra["lik_params"] = [float(rng.uniform(0.1, 2.0)), float(rng.uniform(0.05, 0.95))]
    if fam == "student_t":
        extra["lik_params"] = [float(rng.uniform(0.1, 2.0)), float(rng.uniform(1.0, 30.0))]
    if fam not in ("normal", "categorical", "normal_meanscale") and rng.random() < 0.3:
        extra["offset"] = rng.normal(0, 0.3, n)  # another additive term of the linear predictor
    elif fam in ("categorical", "normal_meanscale") and rng.random() < 0.3:
        extra["offset"] = rng.normal(0, 0.3, (K, n))  # ... of every linear predictor of a K-vector model
    return dict(**extra, name=f"fuzz{seed}", compat=int(compat), response=response, X=X, Y=Y, m=m, P=P, steps=int(rng.integers(2, 5) if large else rng.integers(4, 14)), batch=batch, rules=rules,
                prior=rng.uniform(0.5, 3.0, p), seed=int(rng.integers(0, 2**31)), family=fam, K=K,
                alpha=float(rng.choice([0.95, 0.5, 0.999])), beta=float(rng.choice([2.0, 0.5, 1.0])))
