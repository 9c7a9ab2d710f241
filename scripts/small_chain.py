#!/usr/bin/env python3
"""Real sampler at small sizes: device chain vs the oracle's lean (SciPy) chain on the host, same seed (development)."""
import os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'gaussianprocess-mcmc_amd')); sys.path.insert(0, R)
from gpmc import framework
from oracle import sds_ref
iters = 30
import gpmc
for a in sys.argv[1:]:
    if '=' in a:
        k, v = a.split('=')
        gpmc.set_option(k, int(v))
for n in [int(a) for a in sys.argv[1:] if a.isdigit()] or [200, 455, 1024, 2048]:
    x, y = sds_ref.synthetic_demo_data(n=n, seed=124)
    y = np.clip(y, 1., 99.)
    np.random.seed(5); framework.run_chain(x, y, 2, framework.DEMO_HYP0)
    np.random.seed(5); torch.cuda.synchronize(); t0 = time.perf_counter()
    _, hg, ng = framework.run_chain(x, y, iters, framework.DEMO_HYP0)
    torch.cuda.synchronize(); tg = time.perf_counter() - t0
    np.random.seed(5); t0 = time.perf_counter()
    _, hc, nc = sds_ref.run_chain(x, y, iters, sds_ref.DEMO_HYP0, literal=False)
    tc = time.perf_counter() - t0
    print('n=%5d: device %.1f it/s (%.2f ms per aux_var_model), host lean %.1f it/s; proposals/iter %.1f / %.1f; same decisions %s'
          % (n, iters / tg, 1e3 * tg / (np.sum(ng) + iters), iters / tc, np.mean(ng), np.mean(nc), np.array_equal(ng, nc)), flush=True)
