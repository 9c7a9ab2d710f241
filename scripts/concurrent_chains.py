#!/usr/bin/env python3
"""k independent chains on ONE GPU, each strictly sequential (the next evaluation of a chain is enqueued only after its
previous result has been read, as the sampler must), interleaved by one host thread: whole-GPU evaluations per second.
usage: python3 scripts/concurrent_chains.py N k[,k...] [option=value ...]"""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gaussianprocess-mcmc_amd'))
import gpmc
from gpmc import core
n = int(sys.argv[1]); ks = [int(v) for v in sys.argv[2].split(',')]
for kv in sys.argv[3:]:                                  # options: name=value
    gpmc.set_option(kv.split('=')[0], int(kv.split('=')[1]))
hyps = [[1.0, 10.0, 1.2], [2.0, 8.0, 1.0], [1.5, 9.0, 1.1]]
plans = []
for c in range(max(ks)):
    rng = np.random.RandomState(100 + c)
    X = rng.uniform(0, 10, (n, 2)); g = rng.standard_normal(n) * 5
    p = core.LogMarginalPlan(X, g)
    for j in range(4):
        p(hyps[j % 3])
    plans.append(p)
torch.cuda.synchronize()
for k in ks:
    ps = plans[:k]
    steps = max(8, int(0.5e3 / (0.25 * n / 1024)) // 1)      # ~0.5 s of single-chain work
    for p in ps:
        p.eval_async(hyps[0], 0)
    t0 = time.perf_counter()
    done = 0
    for s in range(steps):
        for p in ps:
            p.result(0)                                   # the chain's decision point: host has the value
            p.eval_async(hyps[(s + 1) % 3], 0)
            done += 1
    for p in ps:
        p.result(0)
    dt = time.perf_counter() - t0
    print('n=%d chains=%d: %.1f evals/s whole GPU (%.3f ms per evaluation per chain)' % (n, k, done / dt, 1e3 * dt * k / done), flush=True)
