"""Wall time of spa(X, k) on the device at the bench shapes (anchor search = k HBM passes of p*n*sizeof(T) each way)."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import conftest  # noqa: F401
import numpy as np
import nmfx

for (p, n, k, T) in [(4096, 4096, 64, np.float32), (16384, 16384, 256, np.float32), (8192, 8192, 128, np.float64)]:
    rng = np.random.default_rng(0)
    W = rng.random((p, k), dtype=np.float32) + 0.1
    H = rng.random((k, n), dtype=np.float32) ** 8
    X = np.asfortranarray((W @ H).astype(T))
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        for ws in (16, 0):
            t0 = time.perf_counter()
            anchors, unsolved = ctx.spa_init(warm_sweeps=ws)
            dt = time.perf_counter() - t0
            Wd = np.empty((p, k), dtype=T, order="F"); Hd = np.empty((k, n), dtype=T, order="F")
            ctx.get_factors(Wd, Hd)
            res = np.linalg.norm(X - Wd @ Hd) / np.linalg.norm(X)
            print(f"spa {T.__name__} {p}x{n} k={k} warm_sweeps={ws}: {dt*1e3:.1f} ms, unsolved {unsolved}, distinct anchors {len(set(anchors.tolist()))}, "
                  f"rel residual {res:.3e}, passes bytes {2*k*p*n*np.dtype(T).itemsize/1e9:.1f} GB", flush=True)
