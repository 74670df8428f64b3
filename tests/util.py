"""Shared helpers for the parity tests."""
import numpy as np
import scipy.sparse as sp


def bsr_to_scipy(rowptr, col, val9):
    nb = rowptr.size - 1
    return sp.bsr_matrix((np.asarray(val9).reshape(-1, 3, 3), col, rowptr), shape=(3 * nb, 3 * nb)).tocsr()


def eqmajor_to_interleaved(v, nc):
    return np.ascontiguousarray(np.asarray(v).reshape(3, nc).T).ravel()


def interleaved_to_eqmajor(v, nc):
    return np.ascontiguousarray(np.asarray(v).reshape(nc, 3).T).ravel()


def random_block_matrix(rowptr, col, seed=1, dominance=6.0):
    """SURVEY 8d 'SpMV micro': blocks = I*(dominance+u) on the diagonal, -0.3*u off it."""
    rng = np.random.default_rng(seed)
    nb = rowptr.size - 1
    val = -0.3 * rng.random((col.size, 9))
    rows = np.repeat(np.arange(nb), np.diff(rowptr))
    d = np.flatnonzero(rows == col)
    val[d] = 0.3 * rng.random((d.size, 9))
    val[d, 0] += dominance + rng.random(d.size); val[d, 4] += dominance + rng.random(d.size); val[d, 8] += dominance + rng.random(d.size)
    return val


def rel_err(a, b):
    return np.abs(np.asarray(a) - np.asarray(b)).max() / (np.abs(np.asarray(b)).max() + 1e-300)
