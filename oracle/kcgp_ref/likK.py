"""Likelihood objects (call sites ``kcMCMC/sliceSample.py:47-50,62,117-118,142-143,279``).

``TruncatedGauss2`` is the reference author's own class and is absent from the
repository: semantics INFERRED from call sites only -- PARITY UNPINNED.
O(N) host arithmetic; not a roofline target.
"""
import numpy as np
from scipy.special import log_ndtr, ndtr, ndtri

_LOG_SQRT_2PI = 0.5 * np.log(2. * np.pi)


def _log_diff_ndtr(b, a):
    """log(Phi(b) - Phi(a)) for b > a, stable in both tails."""
    b = np.asarray(b, dtype=np.float64)
    a = np.asarray(a, dtype=np.float64)
    flip = a > 0                                   # use the upper tail when both are positive
    lo = np.where(flip, -b, a)
    hi = np.where(flip, -a, b)
    lh, ll = log_ndtr(hi), log_ndtr(lo)
    return lh + np.log1p(-np.exp(ll - lh))


class Gauss(object):
    def __init__(self, log_sigma=np.log(0.1)):
        self.sn = float(np.exp(log_sigma))

    def evaluate(self, y=None, mu=None, s2=None):
        if y is None:                              # predictive moments
            v = (0. if s2 is None else np.asarray(s2)) + self.sn ** 2
            sd = np.sqrt(v)
            return mu, mu - 1.96 * sd, mu + 1.96 * sd
        v = self.sn ** 2 + (0. if s2 is None else np.asarray(s2))
        r = np.asarray(y) - np.asarray(mu)
        return float(np.sum(-0.5 * r * r / v - 0.5 * np.log(v) - _LOG_SQRT_2PI))


class TruncatedGauss2(object):
    """Gaussian noise of std ``sn`` truncated to [lower, upper].

    ``evaluate(y=, mu=)``          -> summed log density (scalar)
    ``evaluate(y=, mu=, s2=)``     -> same with variance sn^2 + s2
    ``evaluate(mu=, s2=)``         -> (Ymu, Lower, Upper): truncated mean and the
                                      2.5 % / 97.5 % quantiles of the truncated predictive
    """

    def __init__(self, upper, lower, log_sigma=np.log(0.1)):
        self.upper = upper
        self.lower = lower
        self.sn = float(np.exp(log_sigma))

    def evaluate(self, y=None, mu=None, s2=None):
        mu = np.asarray(mu, dtype=np.float64)
        v = self.sn ** 2 + (0. if s2 is None else np.asarray(s2, dtype=np.float64))
        sd = np.sqrt(v)
        a = (self.lower - mu) / sd
        b = (self.upper - mu) / sd
        logZ = _log_diff_ndtr(b, a)
        if y is not None:
            y = np.asarray(y, dtype=np.float64)
            z = (y - mu) / sd
            lp = -0.5 * z * z - np.log(sd) - _LOG_SQRT_2PI - logZ
            lp = np.where((y < self.lower) | (y > self.upper), -np.inf, lp)
            return float(np.sum(lp))
        Z = np.exp(logZ)
        pa = np.exp(-0.5 * a * a) / np.sqrt(2. * np.pi)
        pb = np.exp(-0.5 * b * b) / np.sqrt(2. * np.pi)
        ymu = mu + sd * (pa - pb) / Z
        Fa = ndtr(a)
        lo = mu + sd * ndtri(Fa + 0.025 * Z)
        up = mu + sd * ndtri(Fa + 0.975 * Z)
        return ymu, lo, up
