"""Round 6: the one Float64 outlier of the ProjectedALS stress sweep (775 x 490, k = 416: 2.1e-2 against the oracle) under the switches
that select the factorisation / solve kernels -- is it conditioning or a kernel?"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import conftest  # noqa: F401
import numpy as np
import nmf_oracle as orc
import nmfx
from problems import planted, rel_trace_err

T, p, n, k, seed = np.float64, 775, 490, 416, 203930532
X, W0, H0 = planted(p, n, k, T, seed=seed, normalize=False, zeroh=True)
oo = orc.Opts(maxiter=6, tol=1e-30, lambda_w=0.1, lambda_h=0.1); oo.track_objective = True
ro = orc.solve("projals", X, W0.copy(order="F"), H0.copy(order="F"), oo)
print("oracle trace", ro.trace)
for env in ({}, {"NMFX_POTRS": "0"}, {"NMFX_CHOL_SLOTS": "0"}, {"NMFX_POTRS_STRIP": "0"}):
    os.environ.update(env)
    W, H = W0.copy(order="F"), H0.copy(order="F")
    r = nmfx.solve(nmfx.ProjectedALS(T, maxiter=6, tol=1e-30, lambda_w=0.1, lambda_h=0.1), X, W, H, track_objective=True)
    for key in env: del os.environ[key]
    print(env, "err", rel_trace_err(r.trace, ro.trace), "per point", np.abs(r.trace - ro.trace) / np.abs(ro.trace))
# conditioning of the Grams along the oracle's path
Wc, Hc = W0.copy(order="F"), H0.copy(order="F")
orc.solve("projals", X, Wc, Hc, orc.Opts(maxiter=1, tol=1e-30, lambda_w=0.1, lambda_h=0.1))
print("cond(H H' + 0.1 I) after one iteration", np.linalg.cond(Hc @ Hc.T + 0.1 * np.eye(k)), "cond(W'W + 0.1 I)", np.linalg.cond(Wc.T @ Wc + 0.1 * np.eye(k)))
