"""NumPy model of the column-sharded formulation the GPU path implements (test infrastructure).

Rank g owns X_g = X[:, c0:c1], H_g; W is replicated between iterations.  Two W-side formulations (include/nmfx.h):
  replicated W update : ONE sum all-reduce of the packed buffer [X_g H_g' | H_g H_g' | rowsum(H_g)] (+ the 2k stop statistics
                        of H), every rank applies the full W update                                   -> step(), step_cd()
  row-sharded W side  : reduce-scatter of X_g H_g' by row blocks (+ all-reduce of the k x k / k-vector tail), rank g updates
                        ITS rows of W, all-gather of the rows; alspgrad's sub-solvers run on the local columns of H / rows of
                        W with all-reduced line-search scalars (one global step size)                 -> step_rows(), step_alspgrad()
The algebra is the Gram form used on the device:  (W'W) H  instead of W'(WH),  W (HH') instead of (WH) H'.
The collectives are injected (`allreduce(np_array) -> np_array`, or a Comm object with reduce_scatter_rows / all_gather_rows;
gloo in the tests)."""
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


# ---------------------------------------------------------------------------------------------------------------------
# Row-sharded W side (csrc/solver_impl.hpp: scatter_w_numerator / gather_w_rows; csrc/alspgrad_impl.hpp: w_subsolve)
# ---------------------------------------------------------------------------------------------------------------------
def row_block(p, rank, world):
    """Rows [r0, r1) of W that `rank` updates (the device uses equal blocks of the padded P; any partition works)."""
    base, rem = divmod(p, world)
    r0 = rank * base + min(rank, rem)
    return r0, r0 + base + (1 if rank < rem else 0)


def step_rows(alg, Xg, W, Hg, lam_w, lam_h, delta, comm, update_H=True):
    """One outer iteration with the row-sharded W side.  comm: .allreduce(a), .reduce_scatter_rows(A) -> summed rows of this
    rank, .all_gather_rows(A_rows) -> full matrix, .row_range(p)."""
    T = Xg.dtype.type
    k = W.shape[1]
    p = W.shape[0]
    r0, r1 = comm.row_range(p)
    preW, preH = W.copy(), Hg.copy()
    I = np.eye(k, dtype=T)
    if alg == "multmse":
        if update_H:
            Hg *= np.maximum(T(0), W.T @ Xg - T(lam_h)) / ((W.T @ W) @ Hg + T(delta))
        num = comm.reduce_scatter_rows(Xg @ Hg.T)
        HHt = comm.allreduce(Hg @ Hg.T)
        Wr = W[r0:r1] * (np.maximum(T(0), num - T(lam_w)) / (W[r0:r1] @ HHt + T(delta)))
    elif alg == "multdiv":
        if update_H:
            Q = Xg / (W @ Hg + T(delta))
            Hg *= (W.T @ Q) / (W.sum(axis=0, dtype=np.float64).astype(T) + T(lam_h))[:, None]
        Q = Xg / (W @ Hg + T(delta))
        num = comm.reduce_scatter_rows(Q @ Hg.T)
        sH = comm.allreduce(Hg.sum(axis=1, dtype=np.float64).astype(T))
        Wr = W[r0:r1] * (num / (sH + T(lam_w))[None, :])
    elif alg == "projals":
        if update_H:
            A = W.T @ W + T(lam_h) * I
            Hg[...] = np.maximum(np.linalg.solve(A.astype(np.float64), (W.T @ Xg).astype(np.float64)), 0).astype(T)
        num = comm.reduce_scatter_rows(Xg @ Hg.T)
        HHt = comm.allreduce(Hg @ Hg.T) + T(lam_w) * I
        Wr = np.maximum(num.astype(np.float64) @ np.linalg.inv(HHt.astype(np.float64)), 0).astype(T)
    else:
        raise ValueError(alg)
    # column statistics of W: partial sums over the rank's rows travel with the all-gather and are added in rank order
    dwp, swp = _stats(Wr, preW[r0:r1], 0)
    W[...] = comm.all_gather_rows(Wr)
    ws = comm.allreduce(np.concatenate([dwp, swp]))
    dh, sh = _stats(Hg, preH, 1)
    hs = comm.allreduce(np.concatenate([dh, sh]))
    return ws[:k], ws[k:], hs[:k], hs[k:]


def pgrad_subsolve_sharded(Z, Gram, B, left, maxiter, traceiter, tolg, beta, sigma, T, allreduce, cnt):
    """_alspgrad_updateh! / _alspgrad_updatew! (src/alspgrad.jl:86-191, 242-347) on a SHARD of Z (columns of H when left,
    rows of W otherwise): the gradient, the trial step and the accept / restore are local, the four scalars of the line search
    (projgradnorm^2, <G,D>, <Gram D,D>, ||Zp - Zn||^2) are summed over the shards -- one global step size, as in the reference."""
    t = 0
    converged = False
    decr_alpha = True
    alpha = T(1)
    beta, sigma = T(beta), T(sigma)
    epsT = T(np.finfo(T).eps)
    Zp = None
    gsum = lambda v: T(allreduce(np.array([float(v)]))[0])   # noqa: E731
    while (not converged) and t < maxiter:
        t += 1
        G = (Gram @ Z if left else Z @ Gram) - B
        m = (G < 0) | (Z > 0)
        pgnrm = T(np.sqrt(gsum(np.sum(np.where(m, G * G, 0), dtype=np.float64))))
        if pgnrm < T(tolg):
            converged = True
        it = 0
        if not converged:
            while it < traceiter:
                it += 1
                cnt["backtracks"] += 1
                Zn = np.maximum(Z - alpha * G, T(0))
                D = Zn - Z
                dv1 = gsum(np.vdot(G, D))
                GD = Gram @ D if left else D @ Gram
                dv2 = gsum(np.vdot(GD, D))
                suff_decr = bool(((T(1) - sigma) * dv1 + T(0.5) * dv2) < 0)
                if it == 1:
                    decr_alpha = not suff_decr
                    Zp = Z.copy()
                if decr_alpha:
                    if suff_decr:
                        Z[...] = Zn
                        break
                    alpha = T(alpha * beta)
                else:
                    approx = bool(T(np.sqrt(gsum(np.sum((Zp - Zn) ** 2, dtype=np.float64)))) <= epsT)
                    if (not suff_decr) or approx:
                        Z[...] = Zp
                        break
                    alpha = T(alpha / beta)
                    Zp = Zn.copy()
        cnt["inner"] += 1
    return t


def step_alspgrad(Xg, W, Hg, state, comm, maxsubiter=200, traceiter=20, beta=0.2, sigma=0.01, update_H=True):
    """update_wh!(::ALSPGradUpd) (src/alspgrad.jl:400-425), sharded: H sub-solve on the local columns, W sub-solve on the
    rank's rows (numerator by reduce-scatter, rows all-gathered afterwards); state = {"tolg": T, "inner": 0, "backtracks": 0}."""
    T = Xg.dtype.type
    p = W.shape[0]
    r0, r1 = comm.row_range(p)
    if update_H:
        itH = pgrad_subsolve_sharded(Hg, W.T @ W, W.T @ Xg, True, maxsubiter, traceiter, state["tolg"], beta, sigma, T, comm.allreduce, state)
        if itH == 1:
            state["tolg"] = T(np.float64(state["tolg"]) * 0.1)
    num = comm.reduce_scatter_rows(Xg @ Hg.T)
    HHt = comm.allreduce(Hg @ Hg.T)
    Wr = W[r0:r1].copy()
    itW = pgrad_subsolve_sharded(Wr, HHt, num, False, maxsubiter, traceiter, state["tolg"], beta, sigma, T, comm.allreduce, state)
    if itW == 1:
        state["tolg"] = T(np.float64(state["tolg"]) * 0.1)
    W[...] = comm.all_gather_rows(Wr)
