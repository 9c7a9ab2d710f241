"""``covK`` mirror: ``Kernel(log-hyp).getCovMatrix(x=, z=, mode=)`` on the MI355X.

Same protocol the reference sampler calls (kcMCMC/sliceSample.py:38-39,104-105,136-137,255,262-263;
demoRegression.py:124-125): constructor takes LOG hyper-parameters, ``getCovMatrix`` returns a fresh
dense matrix -- a ``torch`` device tensor (float64 unless ``dtype`` says otherwise) filled by the
``gpmc_kbuild`` HIP kernel, or, after ``gpmc.set_return('numpy')``, the reference's fresh host ndarray.
"""
import numpy as np
import torch

from . import ops


def _x2d(a, dtype):
    t = ops.to_dev(a, dtype)
    return t.reshape(-1, 1) if t.dim() == 1 else t


class Kernel(object):
    kind = None
    dtype = torch.float64

    def kparams(self, diag_add=0.0):
        """host float64 array [sf2, diag_add, ell...] in the layout gpmc_kbuild reads"""
        raise NotImplementedError

    def getCovMatrix(self, x=None, z=None, mode=None, diag_add=0.0, out=None):
        if mode == 'self_test':                       # only the diagonal of k(z, z): sf2 for both families
            nn = _x2d(z, self.dtype).shape[0]
            return ops.ret(torch.full((nn, 1), float(self.kparams()[0]), dtype=self.dtype, device=ops.device()))
        if mode not in ('train', 'cross'):
            raise ValueError("mode must be 'train', 'cross' or 'self_test'")
        X = _x2d(x, self.dtype)
        Z = None if mode == 'train' else _x2d(z, self.dtype)
        n = X.shape[0]
        m = n if Z is None else Z.shape[0]
        kp = ops.to_dev(self.kparams(diag_add if mode == 'train' else 0.0), self.dtype)
        caller_out = out is not None
        if out is None:
            out = torch.empty((n, m), dtype=self.dtype, device=X.device)
        ops.kbuild(self.kind, X, kp, out, n, m, ops.KB_FULL, Z=Z)
        return out if caller_out else ops.ret(out)


class RBF(Kernel):
    """k(a,b) = sf^2 exp(-|a-b|^2 / (2 ell^2)); ctor (log_ell, log_sigma) as at sliceSample.py:104."""
    kind = 'rbf'

    def __init__(self, log_ell=0., log_sigma=0., dtype=torch.float64):
        self.hyp = [log_ell, log_sigma]
        self.dtype = dtype

    def kparams(self, diag_add=0.0):
        return np.array([np.exp(2. * self.hyp[1]), diag_add, np.exp(self.hyp[0])], dtype=np.float64)


class MaternARD52(Kernel):
    """ARD Matern-5/2 (BASELINE config 5): ctor (array of d log length-scales, log_sigma)."""
    kind = 'matern52ard'

    def __init__(self, log_ell, log_sigma=0., dtype=torch.float64):
        self.hyp = [np.atleast_1d(np.asarray(log_ell, dtype=np.float64)), log_sigma]
        self.dtype = dtype

    def kparams(self, diag_add=0.0):
        return np.concatenate([[np.exp(2. * self.hyp[1]), diag_add], np.exp(self.hyp[0])]).astype(np.float64)


def s_diag_scalar(kii, sn):
    """Sii of aux_var_model (sliceSample.py:184-187) for a constant diagonal kii (= sf2 for both kernels)."""
    kinv = 1. / kii
    v1 = (sn ** 2) ** (-1) + kinv
    return max(1. / (v1 - kinv), 0.)


def kernel_from_hyp(hyp, kernel='rbf', dtype=torch.float64):
    """Sampler convention: hyp = (ell.., sf, sn) on the LINEAR scale -> (Kernel object, sn)."""
    hyp = np.asarray(hyp, dtype=np.float64)
    if kernel == 'rbf':
        return RBF(np.log(hyp[0]), np.log(hyp[1]), dtype=dtype), float(hyp[2])
    if kernel == 'matern52ard':
        return MaternARD52(np.log(hyp[:-2]), np.log(hyp[-2]), dtype=dtype), float(hyp[-1])
    raise ValueError(kernel)
