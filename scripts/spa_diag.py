"""Diagnostic: device spa vs the oracle on the test shapes (H distance, residuals, unsolved columns, cond(W'W))."""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import conftest  # noqa: F401  (paths)
import numpy as np
import nmf_oracle as orc
import nmfx
from test_gpu_spa import near_separable

for T in (np.float64, np.float32):
    for shape in [(300, 260, 40), (64, 1000, 64), (1024, 4096, 32)]:
        for noise in (0.0, 0.02):
            p, n, k = shape
            X = near_separable(p, n, k, T, seed=p + k, noise=noise)
            for ws in (0, 16):
                W, H, info = nmfx.spa(X, k, return_info=True, warm_sweeps=ws)
                Wo, Ho, ao = orc.spa(X, k)
                G = (W.astype(np.float64).T @ W.astype(np.float64))
                d = np.abs(H - Ho)
                j = np.unravel_index(np.argmax(d), d.shape)[1]
                print(T.__name__, shape, noise, "ws", ws, "anch_eq", info["anchors"].tolist() == list(ao), "unsolved", info["unsolved"],
                      "dH %.3e" % (d.max() / np.abs(Ho).max()), "col", j, "suppG", int((H[:, j] > 0).sum()), "suppO", int((Ho[:, j] > 0).sum()),
                      "res %.6e vs %.6e" % (np.linalg.norm(X - W @ H) / np.linalg.norm(X), np.linalg.norm(X - Wo @ Ho) / np.linalg.norm(X)),
                      "cond %.2e" % np.linalg.cond(G), flush=True)
