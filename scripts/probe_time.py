#!/usr/bin/env python3
"""timing-only probe: like one_eval.py but ignores info (for builds/options that break the numerics on purpose)"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gaussianprocess-mcmc_amd'))
import gpmc
from gpmc import core
n = int(sys.argv[1])
for a in sys.argv[2:]:
    k, v = a.split('='); gpmc.set_option(k, int(v))
rng = np.random.RandomState(124)
X = rng.uniform(0, 10, (n, 2)); g = rng.standard_normal(n) * 5
plan = core.LogMarginalPlan(X, g, 'rbf', settle=False)
hyps = [[1.0, 10.0, 1.2], [2.0, 8.0, 1.0]]
for h in hyps * 2:
    plan.eval_async(h, 0); plan.result(0)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(6):
    plan.eval_async(hyps[i % 2], i)
plan.drain(); torch.cuda.synchronize()
print('n=%d %s: %.3f ms/eval (timing only)' % (n, ' '.join(sys.argv[2:]), (time.perf_counter() - t0) / 6 * 1e3))
plan.close()
