"""Wall time of the device front end of nnmf (rsvd + nndsvd) at the bench shapes."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
import conftest  # noqa: F401
import numpy as np
import nmfx

for (p, n, k, T) in [(4096, 4096, 64, np.float32), (16384, 16384, 256, np.float32)]:
    rng = np.random.default_rng(0)
    W = rng.random((p, k), dtype=np.float32)
    H = rng.random((k, n), dtype=np.float32)
    X = np.asfortranarray((W @ H + 0.01 * rng.random((p, n), dtype=np.float32)).astype(T))
    with nmfx.Context(T, p, n, k) as ctx:
        ctx.set_X(X)
        for rep in range(3):
            t0 = time.perf_counter()
            ctx.rsvd(7, download=False)
            t1 = time.perf_counter()
            ctx.nndsvd_init(None, None, None, variant="ar", zeroh=False, seed=7)
            t2 = time.perf_counter()
            print(f"{T.__name__} {p}x{n} k={k}: rsvd {1e3*(t1-t0):.1f} ms, nndsvd {1e3*(t2-t1):.1f} ms", flush=True)
