"""Synthetic workloads of BASELINE.json `configs` (generators: SURVEY.md 8d)."""

from __future__ import annotations

import math

import numpy as np


def _phi(x):
    return 0.5 * (1.0 + np.vectorize(math.erf)(x / math.sqrt(2.0))) if x.size < 4096 else _phi_fast(x)


def _phi_fast(x):
    from scipy.special import ndtr

    return ndtr(x)


def friedman(U: np.ndarray) -> np.ndarray:
    return (10 * np.sin(np.pi * U[:, 0] * U[:, 1]) + 20 * (U[:, 2] - 0.5) ** 2 + 10 * U[:, 3]
            + 5 * U[:, 4])


def cfg1(seed: int = 3415):
    """Friedman-5, n=500 p=5, m=50, 10 particles (CPU plumbing case)."""
    rng = np.random.default_rng(seed)
    X = rng.uniform(0, 1, (500, 5))
    f = friedman(X)
    return dict(X=X, Y=f + rng.normal(0, 1, 500), f=f, m=50, num_particles=10, family="normal",
                name="cfg1: Friedman n=500 p=5 m=50 P=10")


def cfg2(seed: int = 3415, n: int = 100_000, p: int = 50, m: int = 200, num_particles: int = 40):
    """Gaussian regression, n=100k p=50, m=200, 40 particles (the headline config)."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, p))
    f = friedman(_phi_fast(X[:, :5]))
    return dict(X=X, Y=f + rng.standard_normal(n), f=f, m=m, num_particles=num_particles,
                family="normal", name=f"cfg2: Gaussian n={n} p={p} m={m} P={num_particles}")


def bytes_per_tree_update(n: int, rows_touched: float, K: int = 1, s_x: int = 8, s_f: int = 8) -> float:
    """Algorithmic bytes of one tree update:
    2*(3*s_f*K*n) + (4 + s_x + 4 + s_f + 2*K*s_f) * rows_touched."""
    return 2 * (3 * s_f * K * n) + (4 + s_x + 4 + s_f + 2 * K * s_f) * rows_touched


def cfg4(seed: int = 3415, n: int = 1_000_000, p: int = 100, m: int = 200, num_particles: int = 40):
    """Bernoulli-probit classification, n=1M p=100 (X = 800 MB: exceeds the 256 MiB Infinity Cache)."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, p))
    f = friedman(_phi_fast(X[:, :5]))
    f = (f - f.mean()) / 5.0
    Y = (rng.random(n) < _phi_fast(f)).astype(np.float64)
    return dict(X=X, Y=Y, f=f, m=m, num_particles=num_particles, family="bernoulli_probit",
                name=f"cfg4: Bernoulli-probit n={n} p={p} m={m} P={num_particles}")


def cfg5(seed: int = 3415, n: int = 250_000, p: int = 200, K: int = 4, m: int = 100, num_particles: int = 40):
    """Multi-output (shape=(K, n)) BART with a Categorical-softmax likelihood, all ContinuousSplit."""
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, p))
    rows = [X[:, 0], -X[:, 0] + 0.5 * X[:, 1], 1.5 * X[:, 2] * X[:, 3], np.zeros(n)]
    # (K > 4 -- the run-time-K kernels: further logits on the first covariates; K <= 4 is BASELINE's cfg5 unchanged)
    rows += [0.7 * math.cos(k) * X[:, k % 6] + 0.5 * math.sin(k) * X[:, (k + 2) % 7] for k in range(4, K)]
    F = np.stack(rows)[:K]
    pr = np.exp(F - F.max(axis=0))
    pr /= pr.sum(axis=0)
    Y = (rng.random(n)[None, :] > np.cumsum(pr, axis=0)).sum(axis=0).clip(0, K - 1).astype(np.float64)
    return dict(X=X, Y=Y, f=F, m=m, num_particles=num_particles, family="categorical", K=K,
                name=f"cfg5: Categorical-softmax K={K} n={n} p={p} m={m} P={num_particles}")
