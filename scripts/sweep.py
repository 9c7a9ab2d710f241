#!/usr/bin/env python3
"""Development sweep (not the judged bench): time gpmc log-marginal evals across N / options."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gaussianprocess-mcmc_amd'))
import gpmc  # noqa: E402
from gpmc import core  # noqa: E402


def time_plan(plan, hyps, reps):
    for h in hyps[:2]:
        plan(h)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(reps):
        plan.eval_async(hyps[i % len(hyps)], i % 32)
    plan.drain()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def one(n, reps=5):
    rng = np.random.RandomState(124)
    X = rng.uniform(0, 10, (n, 2))
    g = rng.standard_normal(n) * 5
    plan = core.LogMarginalPlan(X, g, 'rbf')
    dt = time_plan(plan, [[1.0, 10.0, 1.2], [2.0, 8.0, 1.0]], reps)
    print('n=%d default options: %.3f ms/eval' % (n, dt * 1e3), flush=True)
    plan.close()


def main():
    if sys.argv[1:2] == ['--diag']:
        for impl in (1, 2, 3):
            gpmc.set_option('diag_impl', impl)
            print('diag_impl', impl)
            for a in sys.argv[2:]:
                one(int(a))
        return
    if sys.argv[1:2] == ['--tiles']:
        for stm, nb, la in ((3072, 256, 1), (100000, 256, 1), (100000, 128, 1), (100000, 512, 1), (100000, 256, 0), (100000, 128, 0)):
            gpmc.set_option('small_tile_m', stm), gpmc.set_option('outer_nb', nb), gpmc.set_option('lookahead', la)
            print('small_tile_m', stm, 'outer_nb', nb, 'lookahead', la)
            for a in sys.argv[2:]:
                one(int(a))
        return
    if sys.argv[1:2] == ['--la']:
        for la, pad, nb in ((-1, -1, 0), (0, 0, 128), (1, 16000, 256), (-1, -1, 0)):
            gpmc.set_option('lookahead', la), gpmc.set_option('gemm_lds_pad', pad), gpmc.set_option('outer_nb', nb)
            print('lookahead', la, 'gemm_lds_pad', pad, 'outer_nb', nb)
            for a in sys.argv[2:]:
                one(int(a))
        return
    if sys.argv[1:2] == ['--pad']:
        # LDS request of the big-tile GEMM under look-ahead: 73,728 B + pad.  > 81,920 B keeps it to one workgroup per
        # CU; <= 163,840 - 80,904 (PanelShared) = 82,936 B lets a panel workgroup share the CU.
        for pad, prio in ((16000, 0), (0, 0), (0, 1), (16000, 1), (9216, 1), (16000, 0)):
            gpmc.set_option('gemm_lds_pad', pad), gpmc.set_option('panel_prio', prio), gpmc.set_option('lookahead', 1)
            gpmc.set_option('outer_nb', 256)
            print('gemm_lds_pad', pad, 'panel_prio', prio)
            for a in sys.argv[2:]:
                one(int(a), 10)
        return
    if sys.argv[1:2] == ['--prio']:
        for prio, depth, lam in ((0, 1, 3072), (1, 1, 3072), (1, 1, 2048), (1, 1, 1024), (1, 2, 3072), (1, 2, 2048), (0, 1, 3072)):
            gpmc.set_option('panel_prio', prio), gpmc.set_option('la_depth', depth), gpmc.set_option('la_min_m', lam)
            print('panel_prio', prio, 'la_depth', depth, 'la_min_m', lam)
            for a in sys.argv[2:]:
                one(int(a), 10)
        return
    if sys.argv[1:2] == ['--depth']:
        # look-ahead depth 1 vs 2 (potrf.hip) across the tail switch point; value printed for a parity eyeball
        for depth, lam, nb in ((1, 3072, 0), (2, 3072, 0), (2, 2048, 0), (2, 4096, 0), (2, 1024, 0), (2, 3072, 384), (1, 3072, 0)):
            gpmc.set_option('la_depth', depth), gpmc.set_option('la_min_m', lam), gpmc.set_option('outer_nb', nb)
            print('la_depth', depth, 'la_min_m', lam, 'outer_nb', nb)
            for a in sys.argv[2:]:
                n = int(a)
                rng = np.random.RandomState(124)
                X = rng.uniform(0, 10, (n, 2)); g = rng.standard_normal(n) * 5
                plan = core.LogMarginalPlan(X, g, 'rbf')
                v = plan([1.0, 10.0, 1.2])
                dt = time_plan(plan, [[1.0, 10.0, 1.2], [2.0, 8.0, 1.0]], 10)
                print('  n=%d: %.3f ms/eval   logml %.9f' % (n, dt * 1e3, v), flush=True)
                plan.close()
        return
    if sys.argv[1:2] == ['--nbla']:
        for nb, lm in ((256, 3072), (384, 3072), (512, 3072), (384, 4608), (512, 4096), (512, 5120), (256, 3072)):
            gpmc.set_option('outer_nb', nb), gpmc.set_option('lookahead', 1), gpmc.set_option('la_min_m', lm)
            print('outer_nb', nb, 'la_min_m', lm)
            for a in sys.argv[2:]:
                one(int(a))
        return
    if sys.argv[1:2] == ['--bk']:
        for v in (0, 1, 0, 1):
            gpmc.set_option('gemm_small_bk32', v)
            print('gemm_small_bk32', v)
            for a in sys.argv[2:]:
                one(int(a))
        return
    if sys.argv[1:2] == ['--panel']:
        for pi in (1, 2, 1, 2):
            gpmc.set_option('panel_impl', pi)
            print('panel_impl', pi)
            for a in sys.argv[2:]:
                one(int(a))
        return
    if sys.argv[1:2] == ['--w8']:
        for w8 in (0, 1, 0, 1):
            gpmc.set_option('gemm_8wave', w8)
            print('gemm_8wave', w8)
            for a in sys.argv[2:]:
                one(int(a))
        return
    if sys.argv[1:2] == ['--tail']:
        for lm in (0, 3072, 4096, 5120, 6144, 7168):
            gpmc.set_option('la_min_m', lm)
            print('la_min_m', lm)
            for a in sys.argv[2:]:
                one(int(a))
        return
    if sys.argv[1:2] == ['--gemm']:
        for gi, stm, nb in ((0, 3072, 128), (1, 3072, 128), (0, 3072, 256), (1, 3072, 256), (1, 3072, 384)):
            gpmc.set_option('gemm_impl', gi), gpmc.set_option('small_tile_m', stm), gpmc.set_option('outer_nb', nb)
            print('gemm_impl', gi, 'small_tile_m', stm, 'outer_nb', nb)
            for a in sys.argv[2:]:
                one(int(a))
        return
    if sys.argv[1:2] == ['--graph']:
        for gr in (0, 1):
            gpmc.set_option('use_graph', gr)
            print('use_graph', gr)
            for a in sys.argv[2:]:
                one(int(a))
        return
    if sys.argv[1:2] == ['--one']:
        for a in sys.argv[2:]:
            one(int(a))
        return
    sizes = [int(a) for a in sys.argv[1:]] or [2048, 4096, 8192]
    rng = np.random.RandomState(124)
    hyps = [[1.0, 10.0, 1.2], [2.0, 8.0, 1.0], [0.7, 12.0, 1.5], [3.0, 5.0, 2.0]]
    for n in sizes:
        X = rng.uniform(0, 10, (n, 2))
        g = rng.standard_normal(n) * 5
        for la, gr, nb, stm in ((0, 0, 256, 3072), (0, 1, 256, 3072), (1, 1, 256, 3072), (1, 1, 128, 3072),
                                (1, 1, 512, 3072), (1, 1, 256, 0), (1, 1, 256, 100000)):
            gpmc.set_option('lookahead', la), gpmc.set_option('use_graph', gr), gpmc.set_option('outer_nb', nb)
            gpmc.set_option('small_tile_m', stm)
            plan = core.LogMarginalPlan(X, g, 'rbf')
            reps = 20 if n <= 4096 else 8
            dt = time_plan(plan, hyps, reps)
            tf = n ** 3 / 3 / dt / 1e12
            print('n=%5d lookahead=%d graph=%d nb=%3d small_m=%6d : %8.3f ms/eval  %7.1f evals/s  %6.2f TF/s' %
                  (n, la, gr, nb, stm, dt * 1e3, 1 / dt, tf), flush=True)
            plan.close()


if __name__ == '__main__':
    main()
