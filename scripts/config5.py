#!/usr/bin/env python3
"""BASELINE config 5 shape on one GPU: ARD Matern-5/2, N=16384, d=8, fp32 (and fp64 for comparison)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gaussianprocess-mcmc_amd'))
from gpmc import core
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 8
rng = np.random.RandomState(124)
X = rng.uniform(0, 10, (n, d)); g = rng.standard_normal(n) * 5
ell = rng.uniform(0.5, 2.0, d) * 3
hyps = [np.concatenate([ell * s, [2.0, 0.7]]) for s in (1.0, 1.1, 0.9, 1.05)]
import gpmc
cases = ((torch.float32, 256, -1), (torch.float32, 256, 1), (torch.float32, 512, 1), (torch.float64, 256, -1))
for dt, nb, la in cases:
    gpmc.set_option('f32_nb', nb), gpmc.set_option('lookahead', la)
    if la == 1 and dt == torch.float32:
        gpmc.set_option('outer_nb', nb)
    else:
        gpmc.set_option('outer_nb', 0)
    plan = core.LogMarginalPlan(X, g, 'matern52ard', dt)
    v = [plan(h) for h in hyps[:2]]
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 6
    for i in range(reps):
        plan.eval_async(hyps[i % 4], i)
    plan.drain(); torch.cuda.synchronize()
    dtm = (time.perf_counter() - t0) / reps
    print('f32_nb=%d lookahead=%d ' % (nb, la) + '%s n=%d d=%d: %.2f ms/eval  %.1f evals/s  %.1f TF/s   logml=%.8g' % (str(dt)[6:], n, d, dtm * 1e3, 1 / dtm, n ** 3 / 3 / dtm / 1e12, v[0]), flush=True)
    plan.close()
