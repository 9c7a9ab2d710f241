#!/usr/bin/env python3
"""tail kernel (csrc/tail.h): agreement with the launch-chain schedule and LAPACK, then timings (development)."""
import os, sys, time
import numpy as np, scipy.linalg, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gaussianprocess-mcmc_amd'))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpmc
from gpmc import core, tools
rng = np.random.RandomState(5)
for n in (64, 65, 200, 455, 1000, 2048, 3000):
    A = rng.standard_normal((n, n)); A = A @ A.T / n + np.eye(n) * 0.5
    Lref = scipy.linalg.cholesky(A, lower=True)
    out = []
    for tm in (0, 2048, 4096):
        gpmc.set_option('tail_m', tm)
        out.append(tools.jitchol(A).cpu().numpy())
    e = [float(np.abs(o - Lref).max() / np.abs(Lref).max()) for o in out]
    print('jitchol n=%4d  max|L-Lref|/max|L|: chain %.2e  tail2048 %.2e  tail4096 %.2e   tail vs chain %.2e' % (n, e[0], e[1], e[2], float(np.abs(out[1] - out[0]).max())), flush=True)
    assert max(e) < 1e-11, e
for n in (2048, 4096, 6144, 8192):
    X = rng.uniform(0, 10, (n, 2)); g = rng.standard_normal(n) * 5
    hyps = [[1.0, 10.0, 1.2], [2.0, 8.0, 1.0]]
    res = {}
    for tm in (0, 1024, 2048, 3072, 4096):
        gpmc.set_option('tail_m', tm)
        plan = core.LogMarginalPlan(X, g, 'rbf')
        v = [plan(h) for h in hyps]
        v = [plan(h) for h in hyps]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(8):
            plan.eval_async(hyps[i % 2], i)
        plan.drain(); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 8 * 1e3
        res[tm] = (ms, v)
        plan.close()
    ref = res[0][1]
    print('n=%5d  ' % n + '  '.join('tail_m=%d: %.3f ms (rel diff %.1e)' % (tm, res[tm][0], abs(res[tm][1][0] - ref[0]) / abs(ref[0])) for tm in res), flush=True)
