#!/usr/bin/env python3
"""sync-each evaluation time (the sampler's use: result read before the next evaluation) at a few sizes (development; GPMC_LIB selects the build)"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gaussianprocess-mcmc_amd'))
import gpmc
from gpmc import core
for kv in sys.argv[2:]:
    gpmc.set_option(kv.split('=')[0], int(kv.split('=')[1]))
hyps = [[1.0, 10.0, 1.2], [2.0, 8.0, 1.0], [1.5, 9.0, 1.1]]
for n in [int(v) for v in sys.argv[1].split(',')]:
    rng = np.random.RandomState(124)
    p = core.LogMarginalPlan(rng.uniform(0, 10, (n, 2)), rng.standard_normal(n) * 5)
    vals = [p(h) for h in hyps]; p(hyps[0])
    best = 1e9
    for rep in range(3):
        steps = max(20, int(150 / (n / 1024.0) ** 2))
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(steps):
            p.eval_async(hyps[i % 3], 0); r = p.result(0)
        best = min(best, 1e3 * (time.perf_counter() - t0) / steps)
    assert r[0] == vals[(steps - 1) % 3]
    print('n=%d sync-each %.4f ms/eval' % (n, best), flush=True)
    p.close()
