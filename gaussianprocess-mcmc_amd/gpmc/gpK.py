"""Minimal ``gpK.GPR`` holder: the attributes ``inf_mcmc`` reads from a model (kcMCMC/sliceSample.py:246-256,
262-265,279; set up at framework.py:231-236).  The optimise/plot methods of kcGP.gpK.GPR are out of scope
(DESIGN.md 7)."""
import numpy as np

from . import covK, likK


class Zero(object):
    def getMean(self, x):
        return np.zeros((np.asarray(x).shape[0], 1))


class Const(object):
    def __init__(self, c=0.):
        self.c = float(c)

    def getMean(self, x):
        return np.full((np.asarray(x).shape[0], 1), self.c)


class GPR(object):
    def __init__(self):
        self.x = self.y = self.xs = None
        self.meanfunc = Zero()
        self.covfunc = covK.RBF()
        self.likfunc = likK.Gauss()

    def setData(self, x, y):
        x = np.asarray(x, dtype=np.float64)
        self.x = x.reshape(-1, 1) if x.ndim == 1 else x
        self.y = np.asarray(y, dtype=np.float64)

    def setPrior(self, mean=None, kernel=None):
        if mean is not None:
            self.meanfunc = mean
        if kernel is not None:
            self.covfunc = kernel

    def setNoise(self, log_sigma):
        self.likfunc = likK.Gauss(log_sigma)
