"""``tools`` mirror: ``jitchol`` / ``solve_chol`` on the MI355X (call sites kcMCMC/sliceSample.py:196,205,257,258).

``jitchol(A)`` returns a fresh lower factor with a zero upper triangle and never mutates ``A`` (the
reference's callers rely on both, sliceSample.py:197,122).  Failure handling follows the pyGPs 1.3.4
schedule the reference pins (requirements.txt:10): on a non-positive pivot require diag > 0, add
``mean(diag) * 1e-9`` jitter, x10 per retry, ``maxtries`` retries, then ``np.linalg.LinAlgError``.
"""
import numpy as np
import torch

from . import ops


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
        return int(self.info.item())


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
        return ops.unpack_lower(ws.W, ws.np_, n)
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
            return ops.unpack_lower(ws.W, ws.np_, n)
        jitter *= 10.
    raise np.linalg.LinAlgError('not positive definite, even with jitter.')
