#!/usr/bin/env python3
"""Which schedule-knob variant misbehaves (development): runs each variant of the bit-equality test on its own."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gaussianprocess-mcmc_amd'))
import gpmc
from gpmc import core
reset = {'tail_park': 1, 'tail_warm': 1, 'tail_workers': 0, 'dag_workers': 0, 'outer_nb': 0, 'dag_ahead': 1,
         'tail_m': 6144, 'tail_acc16': 1, 'tail_prefetch_t': 72, 'dag_prefetch': 0}
cases = ((4200, [{}, {'tail_park': 0}, {'tail_warm': 0}, {'tail_workers': 96}, {'tail_workers': 480}, {'tail_acc16': 0}, {'tail_prefetch_t': 1}]),
         (7300, [{}, {'tail_park': 0}, {'tail_warm': 0}, {'dag_workers': 64}, {'outer_nb': 256}, {'outer_nb': 512}, {'dag_ahead': 0},
                 {'tail_m': 4096}, {'tail_acc16': 0}, {'tail_prefetch_t': 0}, {'tail_prefetch_t': 1, 'dag_prefetch': 1},
                 {'outer_nb': 384}]))
for n, variants in cases:
    rng = np.random.RandomState(31)
    x = rng.uniform(0, 10, (n, 2)); g = np.random.RandomState(32).standard_normal(n) * 5.
    for opts in variants:
        for k, v in {**reset, **opts}.items():
            gpmc.set_option(k, v)
        try:
            plan = core.LogMarginalPlan(x, g, 'rbf', settle=False)
            r = []
            for rep in range(3):
                r.append((plan([1.0, 10.0, 1.2]), plan([2.0, 7.0, 0.9])))
            plan.close()
            print(n, opts, r[0], 'repeatable' if r[0] == r[1] == r[2] else 'NOT REPEATABLE %r' % (r,), flush=True)
        except Exception as e:
            print(n, opts, 'FAILED', str(e)[:120], flush=True)
