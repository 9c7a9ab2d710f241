#!/usr/bin/env python3
"""Option sweep in ONE process (development): python3 scripts/optsweep.py N[,N...] "a=1,b=2" "a=3" ...
Every configuration starts from the library defaults plus the options named `base:` (first argument may be
`base:x=1,y=2`).  Prints ms/eval (12 back-to-back evaluations after 2 warm-ups, best of 2 plans)."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gaussianprocess-mcmc_amd'))
import gpmc
from gpmc import core

args = sys.argv[1:]
sizes = [int(v) for v in args[0].split(',')]
cfgs = args[1:] or ['']
base = {}
if cfgs and cfgs[0].startswith('base:'):
    base = dict(kv.split('=') for kv in cfgs[0][5:].split(',') if kv)
    cfgs = cfgs[1:] or ['']
touched = {}
hyps = [[1.0, 10.0, 1.2], [2.0, 8.0, 1.0]]
dtype = torch.float32 if os.environ.get('F32') == '1' else torch.float64
for n in sizes:
    rng = np.random.RandomState(124)
    X = rng.uniform(0, 10, (n, 2)); g = rng.standard_normal(n) * 5
    for cfg in cfgs:
        opts = dict(base)
        opts.update(dict(kv.split('=') for kv in cfg.split(',') if kv))
        for k in opts:
            touched.setdefault(k, gpmc.get_option(k))
        for k, v in touched.items():
            gpmc.set_option(k, v)
        for k, v in opts.items():
            gpmc.set_option(k, int(v))
        best, val = 1e9, None
        for rep in range(2):
            plan = core.LogMarginalPlan(X, g, 'rbf', dtype)
            vals = [plan(h) for h in hyps + hyps[:1]]      # eager, eager (yardstick), recorded
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(12):
                plan.eval_async(hyps[i % 2], i)
            plan.drain(); torch.cuda.synchronize()
            best = min(best, (time.perf_counter() - t0) / 12 * 1e3)
            plan.close()
            val = vals[0]
        print('n=%d  %-48s %.3f ms/eval   logml %.15g' % (n, cfg or '(defaults)', best, val), flush=True)
for k, v in touched.items():
    gpmc.set_option(k, v)
