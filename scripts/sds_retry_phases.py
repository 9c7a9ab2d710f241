#!/usr/bin/env python3
"""Where a proposal that needs the jitter schedule of chol(R) spends its time, recorded vs eager (development)."""
import os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'gaussianprocess-mcmc_amd')); sys.path.insert(0, R)
import gpmc
from gpmc import sds
for kv in [a for a in sys.argv[1:] if '=' in a]:
    gpmc.set_option(kv.split('=')[0], int(kv.split('=')[1]))
import bench
n = 8192
X, g, pts = bench.synth_inputs(n, 2, threads=16)
f = torch.zeros(n, dtype=torch.float64, device='cuda'); gd = torch.as_tensor(g).cuda()
def timed(fn, name, log):
    def w(*a, **k):
        torch.cuda.synchronize(); t0 = time.perf_counter(); r = fn(*a, **k); torch.cuda.synchronize()
        log.append((name, (time.perf_counter() - t0) * 1e3)); return r
    return w
for record in (True, False):
    ws = sds.SDSWorkspace(X)
    ws.aux_var_model(f, pts[0], gd, need_eta=True); ws.begin(f, gd, ws.eta.clone())
    for j in range(3):
        ws.proposal(pts[j], record=record)
    log = []
    ws._make_R = timed(ws._make_R, 'make_R', log); ws._chol_R = timed(ws._chol_R, 'chol_R', log)
    for j in range(len(pts)):
        del log[:]
        torch.cuda.synchronize(); t0 = time.perf_counter()
        ws.proposal(pts[j], record=record)
        torch.cuda.synchronize(); t = (time.perf_counter() - t0) * 1e3
        if ws._retried:
            print('record=%s point %d: %.2f ms total; inside: %s' % (record, j, t, ', '.join('%s %.2f' % x for x in log)), flush=True)
    ws.close(); del ws
