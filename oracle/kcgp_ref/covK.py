"""Covariance objects: ``Kernel(log-hyp...).getCovMatrix(x=, z=, mode=)``.

Call sites restated: ``kcMCMC/sliceSample.py:38-39,104-105,136-137,255,262-263``,
``demoRegression.py:124-125``.  Hyper-parameters arrive as LOGS.
pyGPs 1.3.4 ``Core/cov.py`` algorithm: scale inputs by 1/ell, squared Euclidean
distance by direct differences (scipy ``cdist``), ``sf2 * exp(-A/2)``.
"""
import numpy as np
from scipy.spatial.distance import cdist


def _as2d(a):
    a = np.asarray(a, dtype=np.float64)
    if a.ndim == 1:
        a = a.reshape(-1, 1)
    return a


class Kernel(object):
    def getCovMatrix(self, x=None, z=None, mode=None):
        raise NotImplementedError

    # scaled pairwise statistic shared by the subclasses
    def _sq(self, x, z, inv_ell):
        xs = _as2d(x) * inv_ell
        zs = xs if z is None else _as2d(z) * inv_ell
        return cdist(xs, zs, 'sqeuclidean')


class RBF(Kernel):
    """k(a,b) = sf^2 exp(-|a-b|^2 / (2 ell^2)); ctor takes log(ell), log(sf)."""

    def __init__(self, log_ell=0., log_sigma=0.):
        self.hyp = [log_ell, log_sigma]

    def getCovMatrix(self, x=None, z=None, mode=None):
        ell = np.exp(self.hyp[0])
        sf2 = np.exp(2. * self.hyp[1])
        if mode == 'self_test':                       # only the diagonal: k(z,z)
            nn = _as2d(z).shape[0]
            A = np.zeros((nn, 1))
        elif mode == 'train':
            A = self._sq(x, None, 1. / ell)
        elif mode == 'cross':                         # (n, ns)
            A = self._sq(x, z, 1. / ell)
        else:
            raise ValueError("mode must be 'train', 'cross' or 'self_test'")
        return sf2 * np.exp(-0.5 * A)


class MaternARD52(Kernel):
    """ARD Matern-5/2 (BASELINE config 5; not in the reference).

    r^2 = sum_k (a_k-b_k)^2/ell_k^2 ; k = sf^2 (1 + sqrt5 r + 5 r^2/3) exp(-sqrt5 r)
    ctor: log_ell = array of d log length-scales, log_sigma = log(sf).
    """

    def __init__(self, log_ell, log_sigma=0.):
        self.hyp = [np.atleast_1d(np.asarray(log_ell, dtype=np.float64)), log_sigma]

    def getCovMatrix(self, x=None, z=None, mode=None):
        inv_ell = np.exp(-self.hyp[0])
        sf2 = np.exp(2. * self.hyp[1])
        if mode == 'self_test':
            return sf2 * np.ones((_as2d(z).shape[0], 1))
        if mode == 'train':
            r2 = self._sq(x, None, inv_ell)
        elif mode == 'cross':
            r2 = self._sq(x, z, inv_ell)
        else:
            raise ValueError("mode must be 'train', 'cross' or 'self_test'")
        t = np.sqrt(5.) * np.sqrt(r2)
        return sf2 * (1. + t + (5. / 3.) * r2) * np.exp(-t)
