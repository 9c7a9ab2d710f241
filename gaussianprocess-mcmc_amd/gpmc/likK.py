"""``likK`` mirror (call sites kcMCMC/sliceSample.py:47-50,117-118,142-143,279).

O(N) host arithmetic on a vector that has to cross to the host anyway (the accept test at
sliceSample.py:154 branches on it); not a GPU kernel and not a roofline target.  ``TruncatedGauss2``
is absent from the reference repository: semantics inferred from its call sites, PARITY UNPINNED.
"""
import numpy as np
from scipy.special import log_ndtr, ndtr, ndtri

_LOG_SQRT_2PI = 0.5 * np.log(2. * np.pi)


def _host(a):
    if hasattr(a, 'detach'):
        return a.detach().cpu().numpy().astype(np.float64, copy=False)
    return np.asarray(a, dtype=np.float64)


def _log_diff_ndtr(b, a):
    flip = a > 0
    lo = np.where(flip, -b, a)
    hi = np.where(flip, -a, b)
    lh, ll = log_ndtr(hi), log_ndtr(lo)
    return lh + np.log1p(-np.exp(ll - lh))


class Gauss(object):
    def __init__(self, log_sigma=np.log(0.1)):
        self.sn = float(np.exp(log_sigma))

    def evaluate(self, y=None, mu=None, s2=None):
        mu = _host(mu)
        v = self.sn ** 2 + (0. if s2 is None else _host(s2))
        if y is None:
            sd = np.sqrt(v)
            return mu, mu - 1.96 * sd, mu + 1.96 * sd
        r = _host(y) - mu
        return float(np.sum(-0.5 * r * r / v - 0.5 * np.log(v) - _LOG_SQRT_2PI))


class TruncatedGauss2(object):
    """Gaussian noise of std ``sn`` truncated to [lower, upper] (ctor: upper, lower, log_sigma)."""

    def __init__(self, upper, lower, log_sigma=np.log(0.1)):
        self.upper = upper
        self.lower = lower
        self.sn = float(np.exp(log_sigma))

    def evaluate(self, y=None, mu=None, s2=None):
        mu = _host(mu)
        v = self.sn ** 2 + (0. if s2 is None else _host(s2))
        sd = np.sqrt(v)
        a = (self.lower - mu) / sd
        b = (self.upper - mu) / sd
        logZ = _log_diff_ndtr(b, a)
        if y is not None:
            y = _host(y)
            z = (y - mu) / sd
            lp = -0.5 * z * z - np.log(sd) - _LOG_SQRT_2PI - logZ
            lp = np.where((y < self.lower) | (y > self.upper), -np.inf, lp)
            return float(np.sum(lp))
        Z = np.exp(logZ)
        pa = np.exp(-0.5 * a * a) / np.sqrt(2. * np.pi)
        pb = np.exp(-0.5 * b * b) / np.sqrt(2. * np.pi)
        ymu = mu + sd * (pa - pb) / Z
        Fa = ndtr(a)
        return ymu, mu + sd * ndtri(Fa + 0.025 * Z), mu + sd * ndtri(Fa + 0.975 * Z)
