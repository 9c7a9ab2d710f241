#!/usr/bin/env python3
"""GEMM-only loop for PMC passes: C = C - P P^T (lower tiles, np = 8192) at K = 256, 6 launches with the plain
tile order then 6 with the XCD-aware order (gemm_xcd_map); rocprofv3 --pmc FETCH_SIZE tells whether the order
changes what leaves the L2s.  Also prints wall time per launch.  (development)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gaussianprocess-mcmc_amd'))
import gpmc
from gpmc import ops
np_, kp = 8192, int(sys.argv[1]) if len(sys.argv) > 1 else 256
P = torch.randn((np_, kp), dtype=torch.float64, device='cuda')
C = torch.randn((np_, np_), dtype=torch.float64, device='cuda')
for xm in (0, 1):
    gpmc.set_option('gemm_xcd_map', xm)
    ops.syrk_diff(C, P, kp, C, np_, np_, 0.0)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5):
        ops.syrk_diff(C, P, kp, C, np_, np_, 0.0)
    torch.cuda.synchronize()
    print('K=%d xcd_map=%d: %.3f ms/launch' % (kp, xm, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
