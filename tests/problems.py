"""Seeded synthetic problems shared by the CPU and GPU tests (SURVEY.md section 8d recipe)."""
import numpy as np


def planted(p, n, k, T, seed=20240910, k0=None, noise=0.01, normalize=True, zeroh=False):
    """Dense X = Wg*Hg + noise*U >= 0 (planted rank k0), W0/H0 ~ U[0,1), W0 columns sum to 1
    like randinit(...; normalize=true) (src/interf.jl:43, src/initialization.jl:6-8)."""
    rng = np.random.default_rng(seed)
    k0 = k if k0 is None else k0
    Wg = rng.random((p, k0))
    Hg = rng.random((k0, n))
    X = np.asfortranarray((Wg @ Hg + noise * rng.random((p, n))).astype(T))
    W0 = rng.random((p, k))
    if normalize:
        W0 /= W0.sum(axis=0, keepdims=True)
    H0 = np.zeros((k, n)) if zeroh else rng.random((k, n))
    return X, np.asfortranarray(W0.astype(T)), np.asfortranarray(H0.astype(T))


def uniform(p, n, k, T, seed=1):
    """X ~ U[0,1) as in BASELINE config 1: nnmf(rand(200,500), 5; ...)."""
    rng = np.random.default_rng(seed)
    X = np.asfortranarray(rng.random((p, n)).astype(T))
    W0 = rng.random((p, k))
    W0 /= W0.sum(axis=0, keepdims=True)
    H0 = rng.random((k, n))
    return X, np.asfortranarray(W0.astype(T)), np.asfortranarray(H0.astype(T))


def rel_trace_err(a, b, floor=0.0):
    """largest relative difference of two objective traces; values below `floor` (the objective's own resolution: an exact fit
    leaves (a few eps * ||X||)^2 or 0 depending on the last bit of the factors) count as equal"""
    a = np.maximum(np.asarray(a, dtype=np.float64), floor)
    b = np.maximum(np.asarray(b, dtype=np.float64), floor)
    return float(np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
