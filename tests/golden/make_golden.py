"""Generate the golden vectors under tests/golden/ (committed; re-run only when the oracle changes).

The reference (Julia) cannot run in the build image, and it ships no golden vectors of its own, so the
fixtures are ORACLE goldens: produced by the NumPy restatement and accepted only if the independent C
restatement agrees on the whole objective trajectory (f64: <=1e-12 relative for the multiplicative updates,
<=1e-9 for projals/alspgrad whose Cholesky / line search amplify rounding; f32: 2e-6 / 1e-3).
Each fixture: X (<= 64 x 96), W0, H0, options -> per-iteration objective, niters, converged, final W, H.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", "..", "oracle"))
sys.path.insert(0, os.path.join(HERE, ".."))
import c_oracle as co
import nmf_oracle as orc
from problems import planted

CASES = {
    # name: (alg, p, n, k, maxiter, extra opts)
    "multmse": ("multmse", 48, 80, 6, 40, dict(lambda_w=1e-4, lambda_h=1e-4)),
    "multdiv": ("multdiv", 48, 80, 6, 40, dict()),
    "projals": ("projals", 64, 96, 5, 25, dict(lambda_w=0.05, lambda_h=0.05)),
    "alspgrad": ("alspgrad", 40, 56, 4, 12, dict()),
    # SURVEY.md section 8f rank 2 (added later; `python make_golden.py cd greedycd` writes only these)
    "cd": ("cd", 48, 80, 6, 20, dict(l1_w=2.5e-4, l2_w=7.5e-4, l1_h=2.5e-4, l2_h=7.5e-4)),
    "greedycd": ("greedycd", 64, 96, 5, 8, dict(lambda_w=1e-4, lambda_h=2e-4)),   # f32 greedy sweeps are chaotic near a fit: short run
}


def main():
    only = set(sys.argv[1:])
    for name, (alg, p, n, k, maxiter, extra) in CASES.items():
        if only and name not in only:
            continue
        for T in (np.float64, np.float32):
            # projals: plain U[0,1) W0 (not column-normalised) keeps W'W + lambda*I well conditioned in f32;
            # with normalised columns the first least-squares step amplifies f32 rounding to ~3e-3 between
            # two CPU implementations of the same algorithm (measured numpy-vs-C), which pins nothing.
            X, W0, H0 = planted(p, n, k, T, seed=42 + p, zeroh=(alg == "projals"), normalize=(alg != "projals"))
            o = orc.Opts(maxiter=maxiter, tol=1e-30, track_objective=True, **extra)
            Wn, Hn = W0.copy(order="F"), H0.copy(order="F")
            rn = orc.solve(alg, X, Wn, Hn, o)
            Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
            rc = co.solve(alg, X, Wc, Hc, o)
            tn, tc = np.array(rn.trace), np.array(rc.trace)
            rel = np.max(np.abs(tn - tc) / np.abs(tn))
            # multiplicative updates are well conditioned; the Cholesky / line-search paths amplify rounding
            lim = {"multmse": (1e-12, 2e-6), "multdiv": (1e-12, 2e-6), "projals": (1e-9, 1e-3),
                   "alspgrad": (1e-9, 1e-3), "cd": (1e-12, 1e-4), "greedycd": (1e-11, 2e-4)}[name][0 if T == np.float64 else 1]
            print(f"{name:9s} {T.__name__}: niters {rn.niters}/{rc.niters} objective rel diff numpy-vs-C {rel:.2e}")
            assert rn.niters == rc.niters and rel < lim, (name, T, rel)
            ro = orc.resolve_opts(orc.ALG_NAMES[alg], T, o)
            np.savez_compressed(os.path.join(HERE, f"{name}_{np.dtype(T).name}.npz"), X=X, W0=W0, H0=H0,
                                trace=tn, niters=rn.niters, converged=rn.converged, W=Wn, H=Hn, objvalue=rn.objvalue,
                                opts=np.array([ro.maxiter, ro.tol, ro.lambda_w, ro.lambda_h, ro.delta, ro.tolg]),
                                opts_cd=np.array([ro.l1_w, ro.l2_w, ro.l1_h, ro.l2_h]), alg=alg)


if __name__ == "__main__":
    main()
