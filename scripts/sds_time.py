#!/usr/bin/env python3
"""ms per SDS proposal, point by point, recorded sequence vs eager launches (development): python3 scripts/sds_time.py [N] [option=value ...]"""
import os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'gaussianprocess-mcmc_amd')); sys.path.insert(0, R)
import gpmc
from gpmc import sds
import bench
for kv in [a for a in sys.argv[2:] if '=' in a]:         # options: name=value
    gpmc.set_option(kv.split('=')[0], int(kv.split('=')[1]))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
X, g, pts = bench.synth_inputs(n, 2, threads=16)
f = torch.zeros(n, dtype=torch.float64, device='cuda'); gd = torch.as_tensor(g).cuda()
for record in (True, False):
    ws = sds.SDSWorkspace(X)
    ws.aux_var_model(f, pts[0], gd, need_eta=True); ws.begin(f, gd, ws.eta.clone())
    for j in range(3):
        ws.proposal(pts[j], record=record)
    ts = []
    for rep in range(2):
        for j in range(len(pts)):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            ws.proposal(pts[j], record=record)
            torch.cuda.synchronize(); ts.append(((time.perf_counter() - t0) * 1e3, ws._retried))
    happy = [t for t, r in ts if not r]; retry = [t for t, r in ts if r]
    print('record=%s: happy path median %.3f ms (min %.3f, n=%d) = %.1f /s; with retry median %.2f ms (n=%d); all points %.1f /s' % (
        record, np.median(happy), min(happy), len(happy), 1e3 / np.median(happy), np.median(retry) if retry else 0, len(retry), 1e3 * len(ts) / sum(t for t, _ in ts)), flush=True)
    ws.close(); del ws
