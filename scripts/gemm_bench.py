#!/usr/bin/env python3
"""Steady-state rate of the two FP64 GEMM kernels through gpmc_syrk_diff (C = D - P P^T, lower tiles)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gaussianprocess-mcmc_amd'))
import gpmc
from gpmc import ops
np_ = 8192
for kp in (128, 256, 1024):
    P = torch.randn((np_, kp), dtype=torch.float64, device='cuda')
    C = torch.randn((np_, np_), dtype=torch.float64, device='cuda')
    for gi, ds in ((1, 0), (1, 1), (1, 0), (1, 1)):
        gpmc.set_option('gemm_impl', gi), gpmc.set_option('gemm_persistent', ds)
        ops.syrk_diff(C, P, kp, C, np_, np_, 0.0)
        torch.cuda.synchronize()
        reps = 5 if kp >= 1024 else 20
        t0 = time.perf_counter()
        for _ in range(reps):
            ops.syrk_diff(C, P, kp, C, np_, np_, 0.0)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        tiles = 64 * 65 // 2
        print('kp=%5d gemm_impl=%d persistent=%d: %8.3f ms  tile-flops %.1f TF/s' % (kp, gi, ds, dt * 1e3, tiles * 128 * 128 * kp * 2 / dt / 1e12), flush=True)
