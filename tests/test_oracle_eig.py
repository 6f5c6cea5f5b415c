"""The oracle's symmetric eigen-solver (Householder + QL, the two stages of Eigen::SelfAdjointEigenSolver used at
marginalization_factor.cpp:275,301) against numpy and against the independent cyclic-Jacobi solver kept for this purpose."""
import ctypes as C

import numpy as np
import pytest

_dp = C.POINTER(C.c_double)


def _eig(lib, name, A):
    n = len(A)
    w = np.zeros(n); V = np.zeros((n, n))
    A = np.ascontiguousarray(A)
    getattr(lib, name)(n, A.ctypes.data_as(_dp), w.ctypes.data_as(_dp), V.ctypes.data_as(_dp))
    return w, V


@pytest.mark.parametrize("n", [1, 2, 7, 15, 70, 130])
def test_sym_eig(oracle, n):
    rng = np.random.default_rng(n)
    B = rng.normal(size=(n, n))
    A = B @ B.T * np.exp(rng.normal(size=(n, 1)))
    A = 0.5 * (A + A.T)
    if n > 20:
        A[:4, :] = 0; A[:, :4] = 0                      # rank deficient, like a prior with unobservable directions
    wn = np.linalg.eigvalsh(A)
    scale = max(np.abs(wn).max(), 1e-300)
    for name in ("orc_sym_eig", "orc_sym_eig_jacobi"):
        w, V = _eig(oracle.lib, name, A)
        assert np.all(np.diff(w) >= 0)                   # ascending, like Eigen
        assert np.abs(w - wn).max() <= 1e-12 * scale
        assert np.abs(V @ np.diag(w) @ V.T - A).max() <= 1e-12 * scale
        assert np.abs(V.T @ V - np.eye(n)).max() <= 1e-12
