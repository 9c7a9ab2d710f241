#!/usr/bin/env python3
"""Largest-size sanity (development): one fp64 log-marginal at N (default 32768; 8.6 GB workspace, exercises the
64-bit index paths) against SciPy on the host, plus timing; and the fp32 plan at the same N."""
import os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'gaussianprocess-mcmc_amd')); sys.path.insert(0, R)
from gpmc import core
from oracle import sds_ref
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
rng = np.random.RandomState(1)
X = rng.uniform(0, 20, (n, 2)); g = rng.standard_normal(n) * 5
hyp = [1.0, 10.0, 1.2]
plan = core.LogMarginalPlan(X, g, 'rbf')
a = plan(hyp)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(3):
    plan.eval_async(hyp, i)
plan.drain(); torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 3
print('n=%d f64: %.1f ms/eval (%.1f TF/s of N^3/3), logml %.9e' % (n, dt * 1e3, n ** 3 / 3 / dt / 1e12, a), flush=True)
plan.close()
p32 = core.LogMarginalPlan(X, g, 'rbf', dtype=torch.float32)
a32 = p32(hyp)
print('n=%d f32: logml %.9e  rel diff to f64 %.2e' % (n, a32, abs(a32 - a) / abs(a)), flush=True)
p32.close()
t0 = time.perf_counter()
b = sds_ref.gp_log_marginal(X, g, hyp)
print('SciPy: %.9e in %.1f s; f64 rel err %.2e' % (b, time.perf_counter() - t0, abs(a - b) / abs(b)), flush=True)
