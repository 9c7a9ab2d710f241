#!/usr/bin/env python3
"""A few evaluations at one size with options from the command line (for rocprofv3 timelines; development).

    python3 scripts/one_eval.py N [f32] [name=value ...]
"""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gaussianprocess-mcmc_amd'))
import gpmc
from gpmc import core
n = int(sys.argv[1])
dtype = torch.float64
for a in sys.argv[2:]:
    if a == 'f32':
        dtype = torch.float32
        continue
    k, v = a.split('=')
    gpmc.set_option(k, int(v))
rng = np.random.RandomState(124)
X = rng.uniform(0, 10, (n, 2)); g = rng.standard_normal(n) * 5
plan = core.LogMarginalPlan(X, g, 'rbf', dtype)
hyps = [[1.0, 10.0, 1.2], [2.0, 8.0, 1.0]]
for h in hyps + hyps[:1]:      # eager, eager (yardstick), recorded
    plan(h)
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(6):
    plan.eval_async(hyps[i % 2], i)
plan.drain(); torch.cuda.synchronize()
print('n=%d %s: %.3f ms/eval' % (n, ' '.join(sys.argv[2:]), (time.perf_counter() - t0) / 6 * 1e3))
plan.close()
