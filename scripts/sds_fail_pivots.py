#!/usr/bin/env python3
"""Where does chol(R + 1e-11 I) fail?  First bad pivot of the first attempt, per bench hyper-parameter point. (development)"""
import os, sys
import numpy as np, torch
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'gaussianprocess-mcmc_amd')); sys.path.insert(0, R)
from gpmc import sds
import bench
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
X, g, pts = bench.synth_inputs(n, 2)
ws = sds.SDSWorkspace(X)
f = torch.zeros(n, dtype=torch.float64, device='cuda'); gd = torch.as_tensor(g).cuda()
for j, h in enumerate(pts):
    ws.f_in.copy_(f); ws.g_in.copy_(gd); ws._set_hyp(h); ws._factor_A(0.0); ws._make_R(); ws._chol_R(0.0)
    print('point %2d: first attempt info = %d of %d' % (j, int(ws.infoR.item()), n), flush=True)
