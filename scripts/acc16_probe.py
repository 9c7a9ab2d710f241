#!/usr/bin/env python3
"""tail_acc16 A/B (development): value, bit-equality with the 4x4x4 form, SciPy parity and ms/eval per size.

    python3 scripts/acc16_probe.py [N ...]
"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gaussianprocess-mcmc_amd'))
sys.path.insert(0, ROOT)
import gpmc
from gpmc import core
from oracle import sds_ref

sizes = [int(a) for a in sys.argv[1:] if a.isdigit()] or [2048, 4096, 6144, 8192]
opts = [a for a in sys.argv[1:] if '=' in a]
for a in opts:
    k, v = a.split('=')
    gpmc.set_option(k, int(v))
hyps = [[1.0, 10.0, 1.2], [2.0, 8.0, 1.0]]
for n in sizes:
    rng = np.random.RandomState(124)
    X = rng.uniform(0, 10, (n, 2)); g = rng.standard_normal(n) * 5
    ref = sds_ref.gp_log_marginal(X, g, hyps[0]) if n <= 8192 else None
    res = {}
    for acc in (0, 1, 0, 1):
        gpmc.set_option('tail_acc16', acc)
        plan = core.LogMarginalPlan(X, g, 'rbf')
        vals = [plan(h) for h in hyps]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(12):
            plan.eval_async(hyps[i % 2], i)
        plan.drain(); torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 12 * 1e3
        plan.close()
        res.setdefault(acc, []).append((vals, ms))
    gpmc.set_option('tail_acc16', 0)
    v0, v1 = res[0][0][0], res[1][0][0]
    print('n=%d  4x4x4 %.3f / %.3f ms   16x16x4 %.3f / %.3f ms   bits equal: %s   rel diff %.2e   vs SciPy: %s / %s' % (
        n, res[0][0][1], res[0][1][1], res[1][0][1], res[1][1][1], v0 == v1,
        max(abs(a - b) / abs(a) for a, b in zip(v0, v1)),
        'n/a' if ref is None else '%.1e' % (abs(v0[0] - ref) / abs(ref)),
        'n/a' if ref is None else '%.1e' % (abs(v1[0] - ref) / abs(ref))), flush=True)
