#!/usr/bin/env python3
"""Accuracy of the device Cholesky / log-marginal vs SciPy at a few sizes (development)."""
import os, sys
import numpy as np, scipy.linalg, torch  # noqa: F401
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'gaussianprocess-mcmc_amd')); sys.path.insert(0, R)
from gpmc import core, tools
from oracle import sds_ref
rng = np.random.RandomState(0)
for n in (1000, 4096):
    Q, _ = np.linalg.qr(rng.standard_normal((n, n))); A = (Q * np.logspace(0, -6, n)) @ Q.T
    L = tools.jitchol(A).cpu().numpy(); Lr = scipy.linalg.cholesky(A, lower=True)
    print('n=%d cond 1e6: max|L-Lref|/max|L| = %.2e   residual %.2e' % (n, np.abs(L - Lr).max() / np.abs(Lr).max(), np.abs(L @ L.T - A).max() / np.abs(A).max()))
for n in (2048, 8192):
    X = rng.uniform(0, 10, (n, 2)); g = rng.standard_normal(n) * 5
    for hyp in ([1.0, 10.0, 1.2], [3.0, 12.0, 0.3]):
        a = core.log_marginal(X, g, hyp); b = sds_ref.gp_log_marginal(X, g, hyp)
        print('n=%d hyp=%s: rel err %.2e' % (n, hyp, abs(a - b) / abs(b)))
