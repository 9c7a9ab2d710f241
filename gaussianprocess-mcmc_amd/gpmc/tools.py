"""``tools`` mirror: ``jitchol`` / ``solve_chol`` on the MI355X (call sites kcMCMC/sliceSample.py:196,205,257,258).

``jitchol(A)`` returns a fresh lower factor with a zero upper triangle and never mutates ``A`` (the
reference's callers rely on both, sliceSample.py:197,122).  Failure handling follows the pyGPs 1.3.4
schedule the reference pins (requirements.txt:10): on a non-positive pivot require diag > 0, add
``mean(diag) * 1e-9`` jitter, x10 per retry, ``maxtries`` retries, then ``np.linalg.LinAlgError``.
"""
import numpy as np
import torch

from . import _lib, ops


class FactorWorkspace(object):
    """Padded (mp x np_) row-major lower-trapezoid the blocked Cholesky works in, plus its scratch."""

    def __init__(self, n, nrhs=0, dtype=torch.float64, dev=None):
        dev = dev or ops.device()
        self.n = int(n)
        self.nrhs = int(nrhs)
        self.np_ = ops.pad(n)
        self.mp = self.np_ + (ops.pad(nrhs) if nrhs > 0 else 0)
        self.dtype = dtype
        self.W = torch.empty((self.mp, self.np_), dtype=dtype, device=dev)
        self.info = torch.zeros(1, dtype=torch.int32, device=dev)
        self.scratch = ops.potrf_scratch(self.np_, dev)
        self.out = torch.zeros(4, dtype=torch.float64, device=dev)

    @property
    def rows(self):
        """the riding rows (nrhs_padded x np_) view"""
        return self.W[self.np_:]

    def factor(self):
        ops.potrf(self.W, self.mp, self.np_, self.info, self.scratch)

    def info_host(self):
        return _lib.check_info(self.info.item())


def jitchol(A, maxtries=5, workspace=None):
    A = ops.to_dev(A, A.dtype if isinstance(A, torch.Tensor) else torch.float64)
    if A.dim() != 2 or A.shape[0] != A.shape[1]:
        raise ValueError('jitchol needs a square matrix')
    n = A.shape[0]
    ws = workspace or FactorWorkspace(n, 0, A.dtype, A.device)
    ops.pack_lower(A, ws.W, ws.np_, 0.0)
    if ws.mp > ws.np_:
        ops.pack_rows(None, ws.W, ws.mp, ws.np_, n)
    ws.factor()
    if ws.info_host() == 0:
        return ops.ret(ops.unpack_lower(ws.W, ws.np_, n))
    diagA = torch.diagonal(A)
    if bool((diagA <= 0).any()):
        raise np.linalg.LinAlgError('not pd: non-positive diagonal elements')
    jitter = float(diagA.mean()) * 1e-9
    for _ in range(maxtries):
        ops.pack_lower(A, ws.W, ws.np_, jitter)
        if ws.mp > ws.np_:
            ops.pack_rows(None, ws.W, ws.mp, ws.np_, n)
        ws.factor()
        if ws.info_host() == 0:
            return ops.ret(ops.unpack_lower(ws.W, ws.np_, n))
        jitter *= 10.
    raise np.linalg.LinAlgError('not positive definite, even with jitter.')


def solve_chol(L, B):
    """X = (L^T L)^-1 B with ``L`` the UPPER factor (hence the ``.T`` at sliceSample.py:257) -- pyGPs'
    ``solve(L, solve(L.T, B))``.  Two MFMA triangular row-solves on the device: the forward half on the lower
    factor L^T, the backward half as a forward solve on its reverse-transpose (gpmc_revt_lower)."""
    U = ops.to_dev(L, torch.float64)
    if U.dim() != 2 or U.shape[0] != U.shape[1]:
        raise ValueError('solve_chol needs a square factor')
    n = U.shape[0]
    Bd = ops.to_dev(B, torch.float64)
    vec = Bd.dim() == 1
    if vec:
        Bd = Bd.reshape(-1, 1)
    if Bd.shape[0] != n:
        raise ValueError('B must have as many rows as the factor')
    m = Bd.shape[1]
    np_ = ops.pad(n)
    mr = (m + 63) // 64 * 64
    dev = U.device
    Wl = torch.empty((np_, np_), dtype=torch.float64, device=dev)
    ops.pack_lower(U.t().contiguous(), Wl, np_, 0.0)          # lower factor L^T with identity padding
    R = torch.zeros((mr, np_), dtype=torch.float64, device=dev)
    R[:m, :n] = Bd.t()
    ops.trsm_rows(Wl, np_, R, mr)                              # rows <- rows (L^T)^-T : forward half
    Wt = torch.empty((np_, np_), dtype=torch.float64, device=dev)
    ops.revt_lower(Wl, np_, Wt)
    Rj = R.flip(1).contiguous()                                # rows J  (data movement only)
    ops.trsm_rows(Wt, np_, Rj, mr)
    X = Rj.flip(1)[:m, :n].t().contiguous()
    return ops.ret(X.reshape(-1) if vec else X)
