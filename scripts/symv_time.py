#!/usr/bin/env python3
"""symv_lower / trmv_lower / trmv_upper timings at N = 8192 against their HBM floors (development)."""
import os, sys
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gaussianprocess-mcmc_amd'))
from gpmc import ops
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
W = torch.randn((n, n), dtype=torch.float64, device='cuda'); x = torch.randn(n, dtype=torch.float64, device='cuda')
y = torch.empty_like(x)
for name, fn, nbytes in (('symv_lower', lambda: ops.symv_lower(W, n, x, y), 8 * n * n),
                         ('trmv_lower', lambda: ops.trmv_lower(W, n, x, None, y), 4 * n * n),
                         ('trmv_upper', lambda: ops.trmv_upper(W, n, x, None, 1.0, y), 4 * n * n)):
    for _ in range(3):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        fn()
    e1.record(); e1.synchronize()
    us = 1e3 * e0.elapsed_time(e1) / 20
    print('%s n=%d: %.1f us, %.2f TB/s of the bytes it reads' % (name, n, us, nbytes / (us * 1e-6) / 1e12), flush=True)
