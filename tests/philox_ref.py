"""NumPy twin of the device generator behind nmfx_randinit (nmf.jl_amd/csrc/frontend_impl.hpp): Philox4x32-10, one call per
matrix element, counter = (global column-major element index lo, hi, stream id, 0), key = the 64-bit seed.
Test infrastructure only."""
import numpy as np

M0, M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
W0, W1 = 0x9E3779B9, 0xBB67AE85
MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(c0, c1, c2, c3, k0, k1):
    """Vectorised over uint32 arrays c0..c3; k0, k1 Python ints.  Returns four uint32 arrays."""
    c0, c1, c2, c3 = (np.asarray(c, dtype=np.uint64) for c in (c0, c1, c2, c3))
    for _ in range(10):
        p0 = M0 * c0
        p1 = M1 * c2
        hi0, lo0 = p0 >> np.uint64(32), p0 & MASK
        hi1, lo1 = p1 >> np.uint64(32), p1 & MASK
        c0, c1, c2, c3 = hi1 ^ c1 ^ np.uint64(k0), lo1, hi0 ^ c3 ^ np.uint64(k1), lo0
        k0 = (k0 + W0) & 0xFFFFFFFF
        k1 = (k1 + W1) & 0xFFFFFFFF
    return tuple(c.astype(np.uint32) for c in (c0, c1, c2, c3))


def rand_matrix(T, rows, cols, seed, stream_id, col_offset=0):
    """rows x cols matrix of U[0,1) exactly as randfill_kernel draws it."""
    j, i = np.meshgrid(np.arange(cols, dtype=np.uint64), np.arange(rows, dtype=np.uint64))
    g = i + (j + np.uint64(col_offset)) * np.uint64(rows)
    w0, w1, _, _ = philox4x32_10(g & MASK, g >> np.uint64(32), np.full(g.shape, stream_id, np.uint64), np.zeros(g.shape, np.uint64),
                                 seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    if np.dtype(T) == np.float32:
        return np.asfortranarray((w0 >> np.uint32(8)).astype(np.float32) * np.float32(1.0 / 16777216.0))
    hi = (w0 >> np.uint32(5)).astype(np.float64)
    lo = (w1 >> np.uint32(6)).astype(np.float64)
    return np.asfortranarray((hi * 67108864.0 + lo) * (1.0 / 9007199254740992.0))


def randinit(T, p, n, k, seed, normalize=False, zeroh=False, h_col_offset=0):
    """randinit(X, k; normalize, zeroh) (src/initialization.jl:4-17) with the device generator."""
    W = rand_matrix(T, p, k, seed, 0)
    if normalize:
        W = np.asfortranarray((W / W.sum(axis=0, dtype=np.float64).astype(T)[None, :]).astype(T))
    H = np.zeros((k, n), dtype=T, order="F") if zeroh else rand_matrix(T, k, n, seed, 1, h_col_offset)
    return W, H


def cd_permutation(k, seed, call):
    """Component order of CoordinateDescent(shuffle = true) for call index `call` = 2*(t-1) + side (include/nmfx.h, nmfx_opts.cd_shuffle):
    the permutation that sorts the keys Philox4x32-10(counter = (i, call, 4, 0), key = seed)[0], i = 0..k-1, ties by index."""
    i = np.arange(k, dtype=np.uint64)
    seed &= 0xFFFFFFFF                                   # the 32-bit field, sign-extended to the 64-bit key like the device does
    key = seed if seed < 2**31 else seed | (0xFFFFFFFF << 32)
    u = philox4x32_10(i, np.full(k, call, np.uint64), np.full(k, 4, np.uint64), np.zeros(k, np.uint64), key & 0xFFFFFFFF, (key >> 32) & 0xFFFFFFFF)[0]
    return np.lexsort((np.arange(k), u))
