#!/usr/bin/env python3
"""Development: run a few full SDS proposals at N (default 8192) for rocprofv3."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gaussianprocess-mcmc_amd'))
from gpmc import sds
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.RandomState(0)
X = rng.uniform(0, 10, (n, 2)); g = torch.as_tensor(rng.standard_normal(n) * 5).cuda()
ws = sds.SDSWorkspace(X)
f = torch.zeros(n, dtype=torch.float64, device='cuda')
ws.aux_var_model(f, [1.0, 10.0, 1.2], g, need_eta=True)
eta = ws.eta.clone()
torch.cuda.synchronize()
t0 = time.perf_counter()
for j in range(3):
    ws.aux_var_model(f, [1.0 + 0.1 * j, 9.0, 1.1], g, need_eta=False)
    ws.propose_f(eta).cpu()
torch.cuda.synchronize()
print('n=%d: %.2f ms per SDS proposal' % (n, (time.perf_counter() - t0) / 3 * 1e3))
