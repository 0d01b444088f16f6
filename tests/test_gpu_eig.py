"""The HBM-resident symmetric eigen-solver of the literal marginalisation (be_linalg.h sym_eig_hbm, round 6) against numpy.linalg.eigh, through the
C ABI's stage harness: the blocks marg_exact = 1 decomposes on the canonical workload are 155 .. 197 wide (15 + the landmarks that start in frame 0) --
too large for LDS -- and positive SEMI-definite with eigenvalues spread over many decades (the 1e-8 cut of marginalization_factor.cpp:281-283 exists
because some directions are unobserved)."""
import ctypes as C

import numpy as np
import pytest

import vio_ct

pytestmark = pytest.mark.gpu


def _eig(P, A):
    L = P.lib()
    L.vio_stage_sym_eig.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.vio_stage_sym_eig.restype = C.c_int
    n = A.shape[0]
    A = np.ascontiguousarray(A, np.float64)
    w = np.zeros(n)
    V = np.zeros((n, n))
    us = np.zeros(4)
    rc = L.vio_stage_sym_eig(n, A.ctypes.data, w.ctypes.data, V.ctypes.data, us.ctypes.data)
    assert rc == 0, rc
    return w, V, us


def _check(A, w, V, tol):
    n = A.shape[0]
    nrm = np.abs(np.linalg.eigvalsh(A)).max()
    assert np.abs(V.T @ V - np.eye(n)).max() < tol, "eigenvectors not orthonormal"
    assert np.abs(V @ np.diag(w) @ V.T - A).max() < tol * nrm, "V diag(w) V^T != A"
    assert np.abs(np.sort(w) - np.linalg.eigvalsh(A)).max() < tol * nrm, "eigenvalues"


@pytest.mark.parametrize("n", [2, 17, 64, 105, 160, 197, 320, 512])
def test_random_symmetric_matrix(P, n):
    rng = np.random.default_rng(100 + n)
    B = rng.standard_normal((n, n))
    A = B + B.T
    w, V, us = _eig(P, A)
    _check(A, w, V, 5e-12)


def test_marginalisation_like_block(P):
    """The shape the literal marginalisation hands over: a dense 15 x 15 pose / speed-bias block bordered by F landmark columns whose own block is diagonal,
    positive semi-definite, spectrum from 1e-9 to 1e6, a few exactly dependent directions.  Beyond the decomposition itself the test forms what
    marginalize() uses it for -- the pseudo-inverse with the eigenvalues <= 1e-8 dropped -- and compares it with numpy's."""
    rng = np.random.default_rng(7)
    F, md = 170, 15
    m = md + F
    J = rng.standard_normal((3 * m, m)) * np.logspace(-4.5, 3, m)[None, :]
    J[:, 3] = J[:, 2]                      # an exactly dependent direction
    J[:, md:] = 0
    A = J.T @ J
    Bc = rng.standard_normal((F, md)) * 1e-2
    d = np.abs(rng.standard_normal(F)) * 50 + 1e-3
    A[md:, :md] = Bc
    A[:md, md:] = Bc.T
    A[md:, md:] = np.diag(d)
    A[:md, :md] += Bc.T @ np.diag(1.0 / d) @ Bc    # keeps the whole matrix positive semi-definite
    A = 0.5 * (A + A.T)
    w, V, us = _eig(P, A)
    _check(A, w, V, 5e-12)
    wn, Vn = np.linalg.eigh(A)
    cut = 1e-8
    pin = (V * np.where(w > cut, 1.0 / np.where(w > cut, w, 1.0), 0.0)) @ V.T
    pin_np = (Vn * np.where(wn > cut, 1.0 / np.where(wn > cut, wn, 1.0), 0.0)) @ Vn.T
    assert (w > cut).sum() == (wn > cut).sum()
    assert np.abs(pin - pin_np).max() < 1e-7 * np.abs(pin_np).max()
    assert us[0] < 60e3, us                   # microseconds: the Jacobi sweeps it replaces took 50 - 100 ms on blocks of this size


def test_diagonal_and_repeated_eigenvalues(P):
    A = np.diag([3.0, 3.0, 3.0, 1.0, 1.0, 0.0, 0.0, 7.0] * 20)
    w, V, _ = _eig(P, A)
    _check(A, w, V, 1e-13)
    n = 150
    Q, _ = np.linalg.qr(np.random.default_rng(3).standard_normal((n, n)))
    lam = np.repeat([5.0, 2.0, 0.0], n // 3)
    A = (Q * lam) @ Q.T
    A = 0.5 * (A + A.T)
    w, V, _ = _eig(P, A)
    _check(A, w, V, 1e-12)
