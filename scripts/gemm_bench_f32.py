#!/usr/bin/env python3
"""Steady-state rate of the f32 GEMM kernel through gpmc_syrk_diff (C = D - P P^T, lower tiles) by K. (development)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gaussianprocess-mcmc_amd'))
from gpmc import ops
np_ = 16384
for kp in (128, 256, 512, 1024, 4096):
    P = torch.randn((np_, kp), dtype=torch.float32, device='cuda')
    C = torch.randn((np_, np_), dtype=torch.float32, device='cuda')
    ops.syrk_diff(C, P, kp, C, np_, np_, 0.0)
    torch.cuda.synchronize()
    reps = 3 if kp >= 1024 else 10
    t0 = time.perf_counter()
    for _ in range(reps):
        ops.syrk_diff(C, P, kp, C, np_, np_, 0.0)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    tiles = 128 * 129 // 2
    print('f32 np=%d K=%5d: %8.3f ms  tile-flops %.1f TF/s' % (np_, kp, dt * 1e3, tiles * 128 * 128 * kp * 2 / dt / 1e12), flush=True)
