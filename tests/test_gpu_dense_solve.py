"""The step's dense solve by itself (vil_debug_dense_solve): the Cholesky factorisation of the reduced pose system and its back substitution, both variants the
library runs, against numpy on matrices of every shape the factorisation distinguishes.

variant 1 is what the one-launch iteration runs: 16-wide panels factored by ONE wave with a matrix row per lane (vil_step.hpp, chol_rowwave: leading 4 x 4 block
redundantly, rows 4 .. 67 in the first panel; a second panel wave past 68 rows; a short last panel), back substitution a column per lane (back_subst_cols: tiles
further left than three folded by the other waves).  variant 0: the look-ahead factorisation with 4-wide panels and the back substitution through inverted
diagonal tiles (the other launch structures).  Sizes: D = 67 and 127 are BASELINE's K = 10 and K = 20 windows (6 K + 7 pose columns); 16 and 19 the smallest
the row-per-lane panels take; 64 / 80 / 96 / 128 have no short last panel (96: the right-hand-side row alone in its tile row); 68 is the first size with a
second panel wave (one row in it), 131 the last it covers; 140 falls back to the look-ahead factorisation in both variants."""
import ctypes as C
import numpy as np
import pytest

from mvil_fusion_amd import lib

pytestmark = pytest.mark.gpu


def _dense(be, A, variant):
    R = A.shape[0]; D = R - 1
    A = np.ascontiguousarray(A, dtype=np.float64)
    L = np.zeros((R, R)); x = np.zeros(D); ok = C.c_int32(-1)
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double))
    be.lib.vil_debug_dense_solve.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(C.c_int32), C.c_int32]
    assert be.lib.vil_debug_dense_solve(be.ctx, D, dp(A), dp(L), dp(x), C.byref(ok), variant) == 0
    return L, x, ok.value


def _system(D, seed, cond=1e6):
    """SPD matrix with a prescribed spread of the spectrum (the reduced pose system of a window is scaled by its Jacobi scaling and damped: moderate), and a rhs."""
    rng = np.random.default_rng(seed)
    Q, _ = np.linalg.qr(rng.standard_normal((D, D)))
    M = (Q * np.geomspace(1.0, cond, D)) @ Q.T
    M = 0.5 * (M + M.T)
    A = np.zeros((D + 1, D + 1)); A[:D, :D] = np.tril(M); A[D, :D] = rng.standard_normal(D) * np.sqrt(cond)
    return M, A


@pytest.mark.parametrize("D", [16, 19, 31, 40, 48, 55, 64, 67, 68, 79, 80, 96, 97, 127, 128, 131, 140])
def test_dense_solve_against_numpy(D):
    be = lib.open_vilsolve()
    M, A = _system(D, 100 + D)
    Lr = np.linalg.cholesky(M); yr = np.linalg.solve(Lr, A[D, :D]); xr = np.linalg.solve(M, A[D, :D])
    out = []
    for variant in (0, 1):
        L, x, ok = _dense(be, A, variant)
        assert ok == 1
        Lf = L[:D, :D]
        assert np.allclose(np.triu(Lf, 1), 0.0)
        assert np.abs(Lf - Lr).max() <= 1e-8 * np.abs(Lr).max()                     # (a Cholesky factor is unique; forward error ~ condition x rounding)
        assert np.abs(Lf @ Lf.T - M).max() <= 1e-13 * np.abs(M).max()                 # backward error: rounding only
        assert np.abs(L[D, :D] - yr).max() <= 1e-8 * np.abs(yr).max()                 # forward substitution: the rhs row
        assert np.abs(x - xr).max() <= 1e-8 * np.abs(xr).max()                        # (condition 1e6)
        assert np.abs(M @ x - A[D, :D]).max() <= 1e-12 * np.abs(M).max() * np.abs(x).max()   # residual: rounding only
        out.append((L, x))
    assert np.abs(out[0][0] - out[1][0]).max() <= 1e-8 * np.abs(Lr).max() and np.abs(out[0][1] - out[1][1]).max() <= 1e-8 * np.abs(xr).max()
    be.close()


@pytest.mark.parametrize("D,bad", [(67, 0), (67, 3), (67, 4), (67, 20), (67, 66), (127, 70), (127, 126), (19, 17)])
def test_a_pivot_that_is_not_positive_is_reported(D, bad):
    """An indefinite matrix (one eigen direction flipped so that pivot `bad` is the first non-positive one): both variants return ok = 0 -- the step then raises
    the damping and solves again (estimator.cpp:1400-1414 through ceres' LM loop), it never uses the factor."""
    be = lib.open_vilsolve()
    M, A = _system(D, 7 + D + bad, cond=1e3)
    Lr = np.linalg.cholesky(M)
    Lb = Lr.copy(); S = np.eye(D); S[bad, bad] = -1.0
    Mb = Lb @ S @ Lb.T                                                              # leading minors positive up to `bad`, then negative
    Ab = A.copy(); Ab[:D, :D] = np.tril(0.5 * (Mb + Mb.T))
    for variant in (0, 1):
        _, _, ok = _dense(be, Ab, variant)
        assert ok == 0
    be.close()


def test_dense_solve_is_reproducible_bit_for_bit():
    be = lib.open_vilsolve()
    for D in (67, 127):
        _, A = _system(D, 5)
        L0, x0, _ = _dense(be, A, 1)
        for _ in range(5):
            L1, x1, _ = _dense(be, A, 1)
            assert np.array_equal(L0, L1) and np.array_equal(x0, x1)
    be.close()
