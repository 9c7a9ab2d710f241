#!/usr/bin/env python3
"""Repeatability of the N=8192 evaluation time across plan re-creations (development)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gaussianprocess-mcmc_amd'))
from gpmc import core
n = 8192
rng = np.random.RandomState(124)
X = rng.uniform(0, 10, (n, 2)); g = rng.standard_normal(n) * 5
hyps = [[1.0, 10.0, 1.2], [2.0, 8.0, 1.0], [0.7, 12.0, 1.5], [3.0, 5.0, 2.0]]
for trial in range(24):
    plan = core.LogMarginalPlan(X, g, 'rbf')
    for h in hyps[:2]:
        plan(h)
    per = []
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(8):
            plan.eval_async(hyps[i % 4], i)
        plan.drain(); torch.cuda.synchronize()
        per.append((time.perf_counter() - t0) / 8 * 1e3)
    print('trial %d: %s ms/eval   (eager ms, first replay ms, recording attempts, form) = %.2f %.2f %d %d' % ((trial, ' '.join('%.2f' % p for p in per)) + plan.record_info()) + '  candidates (ms) %s' % (None if plan.settled is None else '%.2f / %.2f' % plan.settled,), flush=True)
    plan.close()
