"""``jitchol`` / ``solve_chol`` (call sites ``kcMCMC/sliceSample.py:196,205,257,258``).

pyGPs 1.3.4 ``Core/tools.py`` algorithm: LAPACK dpotrf (lower); on failure add
``mean(diag)*1e-9`` jitter, x10 per retry, ``maxtries`` retries, else LinAlgError.
"""
import numpy as np
import scipy.linalg


def jitchol(A, maxtries=5):
    A = np.asarray(A, dtype=np.float64)
    L, info = scipy.linalg.lapack.dpotrf(np.asfortranarray(A), lower=1, clean=1)
    if info == 0:
        return np.ascontiguousarray(L)
    diagA = np.diag(A)
    if np.any(diagA <= 0.):
        raise np.linalg.LinAlgError("not pd: non-positive diagonal elements")
    jitter = diagA.mean() * 1e-9
    for _ in range(maxtries):
        try:
            return np.linalg.cholesky(A + np.eye(A.shape[0]) * jitter)
        except np.linalg.LinAlgError:
            jitter *= 10.
    raise np.linalg.LinAlgError("not positive definite, even with jitter.")


def solve_chol(L, B):
    """X = (L^T L)^{-1} B with L the UPPER factor (hence ``.T`` at sliceSample.py:257)."""
    L = np.asarray(L)
    Y = scipy.linalg.solve_triangular(L, B, trans='T', lower=False)
    return scipy.linalg.solve_triangular(L, Y, lower=False)
