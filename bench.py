#!/usr/bin/env python3
"""bench.py -- GP log-marginal evals/sec (= slice-sample steps/sec) on MI355X.

One "step" = one evaluation of log N(g; 0, K(hyp) + S) at fresh hyper-parameters on the workload
BASELINE.json's metric is quoted on: RBF kernel, N = 8192, d = 2, fp64, synthetic (X, g) resident in HBM
(SURVEY.md 8(d)): K build + S diagonal + blocked Cholesky with g riding + reduction, scalar + info back to
the host.  Evaluations are issued back to back on one stream with distinct hyper-parameters cycled from 16
points drawn from the sampler's bracket; nothing is cached between steps.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 8192] [--d 2] [--no-cpu] [--sds] [--sds-cpu] [--chain K]

--sds adds to "extra" the rate of FULL surrogate-data-slice proposals (K, chol(K+S),
V = L^-1 K, R = K - V^T V, m, chol R, f', log-marginal) on the same inputs; --sds-cpu (implies --sds) also times ONE
proposal in the reference's literal op sequence on the host (oracle restatement; about a minute at N=8192).  Both are
off by default so that the default command launches nothing but the metric's kernels (the committed rocprofv3
kernel stats are of that command).

N > 1 (launched by torch.distributed.run, one rank per GPU): every rank runs its own independent chain-shaped
stream of evaluations on its own GPU (weak scaling, no data-path collective); rank 0 prints ONE JSON line with
the whole-job rate = total evaluations / max-over-ranks time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'gaussianprocess-mcmc_amd'))
sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TF = 78.6      # AMD MI355X spec, FP64 matrix (= vector); measured instruction ceilings in DESIGN.md
HBM_PEAK_GBS = 8000.0


def synth_inputs(n, d, seed=124, truth=(1.0, 10.0, 1.2), npoints=16, scale=(10., 10., 5.)):
    """X ~ U(0,10)^(n x d); g ~ N(0, 25 I) stands in for the GP draw (the cost of an evaluation does not depend
    on g); 16 hyper-parameter points from the sampler's bracket [max(h - U(0,s), 0), +s], floored at 0.05."""
    rng = np.random.RandomState(seed)
    X = rng.uniform(0., 10., (n, d))
    g = rng.standard_normal(n) * 5.0
    h, s = np.asarray(truth), np.asarray(scale)
    pts = []
    for _ in range(npoints):
        lo = np.maximum(h - rng.uniform(0., s), 0.)
        pts.append(np.maximum(rng.uniform(lo, lo + s), 0.05))
    return X, g, np.asarray(pts)


def pmc_traffic(n, kernel_prefix='gpmc::gemm4_nt_kernel'):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes over this same
    command (profiles/r*_bench_n<N>_pmc.json, written by scripts/pmc_summary.py: separate FETCH_SIZE / WRITE_SIZE
    passes, FETCH_SIZE doubled for gfx950, KB -> B); launch-weighted mean over the kernel's tile variants.
    Counters cannot be collected from inside the timed process, hence the committed file; None if there is none."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_n%d_pmc.json' % n)))
    if not files:
        return None, None
    with open(files[-1]) as fh:
        doc = json.load(fh)
    num = den = 0.0
    for name, e in doc.get('kernels', {}).items():
        if name.startswith(kernel_prefix) and 'hbm_bytes_per_launch' in e:
            w = 0.5 * (e['launches_fetch_pass'] + e['launches_write_pass'])
            num += w * e['hbm_bytes_per_launch']
            den += w
    return (num / den if den else None), os.path.relpath(files[-1], ROOT)


def cpu_baseline(X, g, pts, budget_s=20.0):
    """B-lean of BASELINE.md: the oracle's SciPy path (cdist+exp, dpotrf, dtrsv) on the host cores, bounded sample."""
    from oracle import sds_ref
    try:
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info()] + [1])
    except Exception:
        threads = os.cpu_count() or 1
    t0 = time.perf_counter()
    vals, k = [], 0
    while True:
        vals.append(sds_ref.gp_log_marginal(X, g, pts[k % len(pts)]))
        k += 1
        el = time.perf_counter() - t0
        if el > budget_s or k >= 8:
            break
    return dict(value=k / el, unit='evals/s', cores=int(threads), kind='port',
                sample='%d evaluations of the same N=%d workload (first hyper-parameter points), %.1f s' % (k, X.shape[0], el)), vals


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--n', type=int, default=8192)
    ap.add_argument('--d', type=int, default=2)
    ap.add_argument('--no-cpu', action='store_true')
    ap.add_argument('--sds', action='store_true', help='also time full SDS proposals (reported in "extra")')
    ap.add_argument('--sds-cpu', action='store_true', help='also time one literal reference-sequence proposal on the host')
    ap.add_argument('--chain', type=int, default=0, metavar='K',
                    help='also run K iterations of the real sampler (surrogate_slice_sampling) on these inputs: iterations/s')
    a = ap.parse_args()

    import torch
    import torch.distributed as dist
    import gpmc
    from gpmc import _lib, chains, core

    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (the gpmc product path has no CPU fallback)')
    rank, world = chains.init_distributed()          # binds cuda:LOCAL_RANK first, then joins the RCCL group
    dev = torch.device('cuda', torch.cuda.current_device())

    X, g, pts = synth_inputs(a.n, a.d, seed=124 + rank)
    # timed region: the plan's default path (the per-evaluation launch sequence replayed as a hipGraph)
    plan = core.LogMarginalPlan(X, g, 'rbf')

    nccl = world > 1 and dist.get_backend() == 'nccl'

    def barrier():
        if world > 1:
            if nccl:
                dist.barrier(device_ids=[dev.index])
            else:
                dist.barrier()
        torch.cuda.synchronize()

    for i in range(a.warmup):
        plan.eval_async(pts[i % len(pts)], i % 32)
    plan.result((a.warmup - 1) % 32 if a.warmup else 0)

    barrier()
    t0 = time.perf_counter()
    for i in range(a.steps):
        plan.eval_async(pts[i % len(pts)], i % 32)
    last = plan.result((a.steps - 1) % 32)
    barrier()
    dt = time.perf_counter() - t0
    if last[3] != 0:
        raise SystemExit('factorisation reported info=%d' % last[3])

    # roofline pass: the SAME K steps again with eager launches, every launch of the dominant kernel (trailing
    # SYRK update) bracketed by HIP events recorded on the stream it is launched on.  A graph replay exposes no
    # per-kernel boundaries to events, hence the second pass; its wall time is reported as ms_per_step_instrumented.
    plan.close()
    gpmc.set_option('use_graph', 0)
    plan = core.LogMarginalPlan(X, g, 'rbf')
    for i in range(min(a.warmup, 2)):
        plan.eval_async(pts[i % len(pts)], i % 32)
    plan.result(0)
    torch.cuda.synchronize()
    _lib.profile_begin()
    t1 = time.perf_counter()
    for i in range(a.steps):
        plan.eval_async(pts[i % len(pts)], i % 32)
    plan.result((a.steps - 1) % 32)
    torch.cuda.synchronize()
    dt_instr = time.perf_counter() - t1
    k_ms, k_launches, k_flops = _lib.profile_read()

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if (nccl or world == 1) else 'cpu')
    if world > 1:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tmax = float(tmax.item())

    # parity of the timed path against the oracle on this rank's first points (checker, outside the timed region)
    cpu = None
    if rank == 0 and world == 1 and not a.no_cpu:          # the host baseline is timed at N = 1 only
        cpu, vals = cpu_baseline(X, g, pts)
        for j, v in enumerate(vals[:2]):
            got = plan(pts[j])
            if abs(got - v) > 1e-8 * abs(v):
                raise SystemExit('parity failure vs SciPy at point %d: %r vs %r' % (j, got, v))

    extra = {}
    if rank == 0:
        # the O(N^2) kernel build alone, against its HBM roofline (bytes = s N^2 written + s N d read;
        # the lower-only build the plan uses is reported against the same s N^2, and says so)
        from gpmc import ops
        np_ = ops.pad(a.n)
        Wk = torch.empty((np_, np_), dtype=torch.float64, device=dev)
        Xd = torch.as_tensor(X).to(dev)
        kp = torch.tensor([pts[0][1] ** 2, 0.0, pts[0][0]], dtype=torch.float64, device=dev)   # [sf2, Sii, ell]
        kb_bytes = 8.0 * a.n * a.n + 8.0 * a.n * a.d
        for tag, flags in (('full', ops.KB_FULL), ('lower_only', ops.KB_LOWER | ops.KB_PAD_IDENTITY)):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            rc = a.n if tag == 'full' else np_           # the padded build also writes the identity padding
            for _ in range(3):
                ops.kbuild('rbf', Xd, kp, Wk, rc, rc, flags)
            e0.record()
            for _ in range(20):
                ops.kbuild('rbf', Xd, kp, Wk, rc, rc, flags)
            e1.record()
            e1.synchronize()
            us = 1e3 * e0.elapsed_time(e1) / 20
            extra['kbuild_' + tag] = {'us': us, 'algorithmic_bytes': kb_bytes, 'GB_per_s': kb_bytes / (us * 1e-6) / 1e9,
                                      'frac_of_hbm_peak': kb_bytes / (us * 1e-6) / 1e9 / HBM_PEAK_GBS}
        extra['kbuild_lower_only']['note'] = 'writes only the tiles touching the lower triangle; rate quoted against the full s*N^2 bytes'
        del Wk
    if (a.sds or a.sds_cpu) and rank == 0:
        from gpmc import sds
        ws = sds.SDSWorkspace(X)
        f = torch.zeros(a.n, dtype=torch.float64, device=dev)
        gd = torch.as_tensor(g).to(dev)
        ws.aux_var_model(f, pts[0], gd, need_eta=True)
        eta = ws.eta.clone()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        reps = 5
        for j in range(reps):
            ws.aux_var_model(f, pts[j % len(pts)], gd, need_eta=False)
            ws.propose_f(eta).cpu()
        torch.cuda.synchronize()
        extra['sds_proposals_per_s'] = reps / (time.perf_counter() - t1)
        extra['sds_flops_per_proposal'] = 4.0 * a.n ** 3 / 3.0     # inverse form (gpmc/sds.py)
        del ws
        if a.sds_cpu:
            # the reference's literal op sequence (sliceSample.py:136-147,165-207) through the oracle restatement
            from oracle import sds_ref
            from oracle.kcgp_ref import covK as ref_covK
            h = pts[0]
            t1 = time.perf_counter()
            Kc = ref_covK.RBF(np.log(h[0]), np.log(h[1])).getCovMatrix(x=X, mode='train')
            _, K_S, m_, C_, L_ = sds_ref.aux_var_model_literal(np.zeros(a.n), Kc, h[2], g)
            _ = np.dot(C_, eta.cpu().numpy()) + m_
            _ = sds_ref.log_marginal_literal(g, K_S, L_)
            extra['sds_cpu_literal_s_per_proposal'] = time.perf_counter() - t1

    if a.chain > 0 and rank == 0:
        # chain-level cross-check: the real sampler from the framework's initial point; targets are a
        # bounded series in the reference's data range, one aux_var_model + several full SDS proposals per iteration
        from gpmc import framework
        yb = np.clip(50. + g, 1., 99.)
        np.random.seed(124)
        framework.run_chain(X, yb, 1, framework.FRAMEWORK_HYP0)                      # warm-up (workspaces)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        _, hh, ntries = framework.run_chain(X, yb, a.chain, framework.FRAMEWORK_HYP0)
        torch.cuda.synchronize()
        el = time.perf_counter() - t1
        extra['chain'] = {'iterations': a.chain, 'iterations_per_s': a.chain / el,
                          'proposals_per_iteration': float(np.mean(ntries)), 'last_hyp': [float(v) for v in hh[:, -1]]}

    if rank == 0:
        value = world * a.steps / tmax
        achieved = (k_flops / (k_ms * 1e-3)) / 1e12 if k_ms > 0 else 0.0
        traffic, traffic_src = pmc_traffic(a.n) if a.d == 2 else (None, None)
        line = {
            'metric': 'GP log-marginal evals/sec (= slice-sample steps/sec) at N=%d' % a.n,
            'value': value, 'unit': 'evals/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': 1e3 * tmax / a.steps, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': 'RBF kernel, N=%d, d=%d, fp64, 1 chain per GPU (BASELINE.json configs[2])' % (a.n, a.d),
                       'global_batch': world, 'parallelism': 'chain-parallel x%d (no data-path collective)' % world,
                       'schedule': 'auto (look-ahead NB=256 while >2048 columns remain, then serial NB=128)',
                       'launch': 'hipGraph replay'},
            'roofline': {'bound': 'mfma', 'kernel': 'gpmc::gemm4_nt_kernel (trailing SYRK update of the blocked Cholesky)',
                         'achieved': achieved, 'peak': FP64_MATRIX_PEAK_TF, 'unit': 'TFLOP/s',
                         'frac': achieved / FP64_MATRIX_PEAK_TF,
                         'launches_per_step': k_launches / max(a.steps, 1),
                         'avg_launch_us': 1e3 * k_ms / max(k_launches, 1),
                         'algorithmic_flops_per_launch': k_flops / max(k_launches, 1),
                         'whole_step_tflops': (a.n ** 3 / 3.0) / (tmax / a.steps) / 1e12,
                         'measured_in': 'second pass over the same steps, eager launches, HIP events on the launch stream',
                         'ms_per_step_instrumented': 1e3 * dt_instr / a.steps,
                         'traffic': traffic, 'traffic_unit': 'HBM-side bytes per launch (PMC)',
                         'traffic_source': traffic_src},
            'cpu_baseline': cpu,
        }
        if extra:
            line['extra'] = extra
        print(json.dumps(line), flush=True)
    plan.close()
    if world > 1:
        barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
