"""NumPy model of the column-sharded formulation the GPU path implements (test infrastructure).

Rank g owns X_g = X[:, c0:c1], H_g; W is replicated.  Per outer iteration ONE sum all-reduce of the packed
buffer [X_g H_g' | H_g H_g' | rowsum(H_g)] (+ the 2k stop statistics of H) -- see nmfx/dist.py.
The algebra is the Gram form used on the device:  (W'W) H  instead of W'(WH),  W (HH') instead of (WH) H'.
`allreduce(np_array) -> np_array` is injected (gloo in the tests, identity for world = 1)."""
import numpy as np


def _stats(new, old, axis):
    d = ((new - old) ** 2).sum(axis=axis, dtype=np.float64)
    s = ((new + old) ** 2).sum(axis=axis, dtype=np.float64)
    return d, s


def _cd_rows(Wv, P, Z):
    """cd sweep of every sample row (rows of Wv), components in order (coorddesc.jl:133-156)."""
    T = Wv.dtype.type
    for t in range(P.shape[0]):
        grad = Wv @ P[t] - Z[:, t]
        if P[t, t] != 0:
            Wv[:, t] = np.maximum(Wv[:, t] - grad / P[t, t], T(0))


def _greedy_rows(Wv, P, G, lam, allmax):
    """greedy sweep of every sample row (greedycd.jl:117-158); `allmax` reduces p_init over the ranks that share the rows."""
    T = Wv.dtype.type
    k = P.shape[0]
    e = T(np.finfo(T).eps)
    prr = np.diag(P).copy()
    G = G + T(lam) if lam > 0 else G.copy()

    def sd(w, g):
        s = np.maximum(T(0), w - g / (e + prr)) - w
        return s, -g * s - (T(0.5) * prr) * (s * s)

    S, D = sd(Wv, G)
    q = np.argmax(D, axis=1)
    p_init = allmax(max(T(-1.0), D[np.arange(len(q)), q].max()))
    out = Wv.copy()
    for i in range(Wv.shape[0]):
        w, g = Wv[i].copy(), G[i].copy()
        s, d = S[i].copy(), D[i].copy()
        qi = int(q[i])
        wn = np.zeros(k, T)
        for _ in range(k * k):
            if d[qi] < T(0.001) * p_init:
                break
            wn[qi] += s[qi]
            g = g + s[qi] * P[qi]
            s, d = sd(w, g)
            qi = int(np.argmax(d))
        out[i] = np.maximum(w + wn, T(0))
    Wv[...] = out


def step_cd(alg, Xg, W, Hg, lam_w, lam_h, allreduce, allmax, update_H=True):
    """CoordinateDescent (no regularisation) / GreedyCD: W FIRST from the all-reduced [X_g H_g' | H_g H_g'], then the local
    H columns from the new W; GreedyCD's H-side p_init is a max over ALL columns -> one max all-reduce."""
    T = Xg.dtype.type
    k = W.shape[1]
    preW, preH = W.copy(), Hg.copy()
    pack = allreduce(np.concatenate([(Xg @ Hg.T).ravel(order="F"), (Hg @ Hg.T).ravel(order="F")]))
    XHt = pack[: W.size].reshape(W.shape, order="F")
    HHt = pack[W.size:].reshape((k, k), order="F")
    if alg == "cd":
        _cd_rows(W, HHt, XHt)
    else:
        _greedy_rows(W, HHt, W @ HHt - XHt, lam_w, lambda v: v)          # rows of W are replicated: local max is global
    if update_H:
        WtW, XtW = W.T @ W, Xg.T @ W
        Ht = Hg.T                                                         # view
        if alg == "cd":
            _cd_rows(Ht, WtW, XtW)
        else:
            _greedy_rows(Ht, WtW, Ht @ WtW - XtW, lam_h, allmax)
    dh, sh = _stats(Hg, preH, 1)
    hs = allreduce(np.concatenate([dh, sh]))
    dw, sw = _stats(W, preW, 0)
    return dw, sw, hs[:k], hs[k:]


def step(alg, Xg, W, Hg, lam_w, lam_h, delta, allreduce, update_H=True):
    T = Xg.dtype.type
    k = W.shape[1]
    preW, preH = W.copy(), Hg.copy()
    if alg == "multmse":
        if update_H:
            Hg *= np.maximum(T(0), W.T @ Xg - T(lam_h)) / ((W.T @ W) @ Hg + T(delta))
        pack = np.concatenate([(Xg @ Hg.T).ravel(order="F"), (Hg @ Hg.T).ravel(order="F"), np.zeros(k, T)])
        pack = allreduce(pack)
        XHt = pack[: W.size].reshape(W.shape, order="F")
        HHt = pack[W.size: W.size + k * k].reshape((k, k), order="F")
        W *= np.maximum(T(0), XHt - T(lam_w)) / (W @ HHt + T(delta))
    elif alg == "multdiv":
        if update_H:
            Q = Xg / (W @ Hg + T(delta))
            Hg *= (W.T @ Q) / (W.sum(axis=0, dtype=np.float64).astype(T) + T(lam_h))[:, None]
        Q = Xg / (W @ Hg + T(delta))
        pack = np.concatenate([(Q @ Hg.T).ravel(order="F"), np.zeros(k * k, T), Hg.sum(axis=1, dtype=np.float64).astype(T)])
        pack = allreduce(pack)
        QHt = pack[: W.size].reshape(W.shape, order="F")
        sH = pack[W.size + k * k:]
        W *= QHt / (sH + T(lam_w))[None, :]
    elif alg == "projals":
        if update_H:
            A = W.T @ W + T(lam_h) * np.eye(k, dtype=T)
            Hg[...] = np.maximum(np.linalg.solve(A.astype(np.float64), (W.T @ Xg).astype(np.float64)), 0).astype(T)
        pack = np.concatenate([(Xg @ Hg.T).ravel(order="F"), (Hg @ Hg.T).ravel(order="F"), np.zeros(k, T)])
        pack = allreduce(pack)
        XHt = pack[: W.size].reshape(W.shape, order="F")
        HHt = pack[W.size: W.size + k * k].reshape((k, k), order="F") + T(lam_w) * np.eye(k, dtype=T)
        W[...] = np.maximum(XHt.astype(np.float64) @ np.linalg.inv(HHt.astype(np.float64)), 0).astype(T)
    else:
        raise ValueError(alg)
    dh, sh = _stats(Hg, preH, 1)
    hs = allreduce(np.concatenate([dh, sh]))
    dw, sw = _stats(W, preW, 0)
    return dw, sw, hs[:k], hs[k:]


def objective(alg, Xg, W, Hg, allreduce):
    WH = W @ Hg
    if alg == "multdiv":
        pos = Xg > 0
        t = np.where(pos, Xg * np.log(np.where(pos, Xg, 1) / WH) - Xg + WH, WH)
        return float(allreduce(np.array([t.sum(dtype=np.float64)]))[0])
    return 0.5 * float(allreduce(np.array([((Xg - WH) ** 2).sum(dtype=np.float64)]))[0])
