#!/usr/bin/env python3
"""GEMM alone (C = C - P P^T, lower tiles of 8192^2, f64) with the second workgroup slot of every CU delayed at
kernel start (gemm_desync = 2 r: blocks 256..511 sleep r x 3.4 us) -- does anti-phasing co-resident workgroups hide
the C traffic?  (development)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gaussianprocess-mcmc_amd'))
import gpmc
from gpmc import ops
np_ = 8192
for kp in (128, 256):
    P = torch.randn((np_, kp), dtype=torch.float64, device='cuda')
    C = torch.randn((np_, np_), dtype=torch.float64, device='cuda')
    for it in range(3):                                      # warm-up: clocks and caches settle
        ops.syrk_diff(C, P, kp, C, np_, np_, 0.0)
    cases = [0] + [256 * b + 2 * r for b in (3, 4, 5, 6, 7, 8) for r in (4, 7)] + [0]
    for ds in cases:
        gpmc.set_option('gemm_desync', ds)
        ops.syrk_diff(C, P, kp, C, np_, np_, 0.0)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            ops.syrk_diff(C, P, kp, C, np_, np_, 0.0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        print('K=%d select bit %d, delay %.1f us: %.3f ms' % (kp, ds >> 8, ((ds & 0xe) >> 1) * 3.4, dt * 1e3), flush=True)
