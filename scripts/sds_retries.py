#!/usr/bin/env python3
"""How often does chol(R + 1e-11 I) need the jitter schedule on the bench's hyper-parameter points? (development)"""
import os, sys, time
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'gaussianprocess-mcmc_amd')); sys.path.insert(0, R)
from gpmc import sds
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
X, g, pts = bench.synth_inputs(n, 2)
ws = sds.SDSWorkspace(X)
calls = {'A': 0, 'R': 0, 'C': 0}
fa, fr, fc = ws._factor_A, ws._make_R, ws._chol_R
def cA(*a): calls['A'] += 1; return fa(*a)
def cR(*a): calls['R'] += 1; return fr(*a)
def cC(*a): calls['C'] += 1; return fc(*a)
ws._factor_A, ws._make_R, ws._chol_R = cA, cR, cC
f = torch.zeros(n, dtype=torch.float64, device='cuda'); gd = torch.as_tensor(g).cuda()
for j, h in enumerate(pts):
    calls['A'] = calls['R'] = calls['C'] = 0
    torch.cuda.synchronize(); t0 = time.perf_counter()
    ws.aux_var_model(f, h, gd, need_eta=False)
    torch.cuda.synchronize()
    print('point %2d hyp=%s: %.1f ms, factor_A x%d, make_R x%d, chol_R x%d' % (j, np.round(h, 3), (time.perf_counter() - t0) * 1e3, calls['A'], calls['R'], calls['C']), flush=True)
