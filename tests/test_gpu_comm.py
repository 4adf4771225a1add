"""RCCL path on the 1-GPU box: a communicator of size 1 makes every ncclAllReduce of the sharded code path run
(packed W-side buffer, H statistics, objective, alspgrad scalars).  Results must equal the no-communicator run
bit for bit (sum over one rank is the identity)."""
import numpy as np
import pytest

import nmfx
from problems import planted

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("alg_name", ["multmse", "multdiv", "projals", "alspgrad"])
def test_nranks1_comm_is_identity(built, alg_name):
    T = np.float32
    p, n, k = 260, 300, 6
    X, W0, H0 = planted(p, n, k, T, seed=5, normalize=(alg_name != "projals"))
    if alg_name in ("multmse", "multdiv"):
        alg = nmfx.MultUpdate(T, obj=alg_name[4:], maxiter=12, tol=1e-30)
    elif alg_name == "projals":
        alg = nmfx.ProjectedALS(T, maxiter=8, tol=1e-30, lambda_w=0.05, lambda_h=0.05)
    else:
        alg = nmfx.ALSPGrad(T, maxiter=4, tol=1e-30)
    outs = []
    for use_comm in (False, True):
        W, H = W0.copy(order="F"), H0.copy(order="F")
        with nmfx.Context(T, p, n, k) as ctx:
            ctx.set_X(X)
            if use_comm:
                ctx.comm_init(nmfx.comm_unique_id(), 0, 1)
            r = nmfx.solve(alg, X, W, H, ctx=ctx, track_objective=True)
        outs.append((W, H, r))
    (W0_, H0_, r0), (W1, H1, r1) = outs
    assert r0.niters == r1.niters
    assert np.array_equal(r0.trace, r1.trace)
    assert np.array_equal(W0_, W1) and np.array_equal(H0_, H1)


def test_leading_dimension_limit(built):
    """The fused epilogues address a wave tile with 32-bit byte offsets: 256 * ld * sizeof(T) must stay below 2^32
     -- a larger p is refused up front with NMFX_ERR_UNSUPPORTED, before anything is allocated."""
    with pytest.raises(nmfx.NMFXError, match="too large"):
        nmfx.Context(np.float32, 4_194_304, 2, 2)
    with pytest.raises(nmfx.NMFXError, match="too large"):
        nmfx.Context(np.float64, 2_097_152, 2, 2)
