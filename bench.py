#!/usr/bin/env python3
"""bench.py -- GP log-marginal evals/sec on MI355X (BASELINE.json's metric).

One "step" = ONE evaluation of log N(g; 0, K(hyp) + S) at fresh hyper-parameters = what the slice sampler
re-evaluates for one hyper-parameter proposal's log-marginal term (kcMCMC/sliceSample.py:136-137,196,147): K build
+ S diagonal + blocked Cholesky with g riding + reduction, N^3/3 flops, scalar + info back on the host.  It is NOT a
full SDS proposal (4 N^3/3 here, two factorisations) nor a sampler iteration (several proposals): those rates are the
`--full` extras `sds_proposals_per_s` and `chain.iterations_per_s`.  Workload: RBF kernel, N = 8192, d = 2, fp64,
synthetic (X, g) resident in HBM (SURVEY.md 8(d) / BASELINE.md 2), hyper-parameters cycled over 16 points drawn from
the sampler's bracket; nothing is cached between steps.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--n 8192] [--d 2] [--pipelined] [--no-cpu] [--no-cpu-ref]
                    [--metric-only] [--full | --sds --sds-cpu --chain K]

Timed region: K steps, each one read back (logml, info) BEFORE the next is enqueued -- the way the sampler consumes
them (sliceSample.py:154 branches on every value).  `--pipelined` times the back-to-back form instead; the default run
reports the other form's rate in `extra.pipelined_evals_per_s`.

--gpus N > 1: when not already under torch.distributed.run, this script spawns N ranks of itself (one per GPU,
LOCAL_RANK = 0..N-1, rendez-vous on 127.0.0.1) BEFORE anything touches the GPU in the parent, and forwards rank 0's
JSON line.  Every rank runs its own independent stream of evaluations on its own GPU with its own inputs (seed
124 + rank): weak scaling, no data-path collective; value = N * K / max-over-ranks time.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, 'gaussianprocess-mcmc_amd'))
sys.path.insert(0, ROOT)

FP64_MATRIX_PEAK_TF = 78.6      # AMD MI355X spec, FP64 matrix (= vector); measured instruction ceilings in DESIGN.md
HBM_PEAK_GBS = 8000.0
RING = 32                       # result slots the timed loops cycle through (the plan has 64)


# ------------------------------------------------------------------------------------------ multi-rank launch
def spawn_ranks(n, argv):
    """`python bench.py --gpus N` without a launcher: start N ranks of this script, one per GPU.  The parent never
    initialises the GPU (it imports neither torch nor gpmc); it forwards rank 0's stdout and fails if any rank fails."""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    out0, _ = procs[0].communicate()
    rcs = [procs[0].returncode]
    for p in procs[1:]:
        try:
            rcs.append(p.wait(timeout=120))
        except subprocess.TimeoutExpired:       # rank 0 is done (or dead) and this one is not: stop it by PID
            p.kill()
            rcs.append(p.wait())
    sys.stdout.write(out0.decode())
    sys.stdout.flush()
    bad = [(r, c) for r, c in enumerate(rcs) if c != 0]
    if bad:
        raise SystemExit('bench.py: ranks failed (rank, exit code): %r' % bad)


# ------------------------------------------------------------------------------------------ inputs
def effective_cpus():
    """cores this process may really use: min(affinity mask, cgroup cpu.max quota) -- the GPU boxes expose 128 logical
    CPUs but grant a 1-GPU job a share of 16, and a BLAS pool sized to 128 there runs ~10x slower than one sized 16."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with open('/sys/fs/cgroup/cpu.max') as fh:
            q, p = fh.read().split()
        if q != 'max':
            n = min(n, max(1, int(round(int(q) / int(p)))))
    except Exception:
        pass
    return max(1, n)


def synth_inputs(n, d, seed=124, truth=(1.0, 10.0, 1.2), npoints=16, scale=(10., 10., 5.), threads=None):
    """RandomState(124 + rank); X ~ U(0,10)^(n x d); truth (ll, sf, sn) = (1, 10, 1.2);
    g = chol(K + sn^2 I) z (host SciPy, outside every timed region); 16 hyper-parameter points from the sampler's
    bracket [max(h - U(0,s), 0), +s] (framework.py:69 scale), floored at 0.05."""
    import scipy.linalg
    from scipy.spatial.distance import cdist
    from threadpoolctl import threadpool_limits
    rng = np.random.RandomState(seed)
    X = rng.uniform(0., 10., (n, d))
    z = rng.standard_normal(n)
    with threadpool_limits(limits=threads):
        K = cdist(X / truth[0], X / truth[0], 'sqeuclidean')
        K *= -0.5
        np.exp(K, out=K)
        K *= truth[1] ** 2
        K[np.diag_indices(n)] += truth[2] ** 2
        L = scipy.linalg.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
        g = L @ z
    del K, L
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


def sq_counters(n, kernel_prefix):
    """MFMA-pipe busy %, issue-stall %, parked %, LDS bank-conflict % and L2 hit % of a kernel from the committed
    rocprofv3 SQ / GRBM / TCC passes over this same command (profiles/r*_bench_n<N>_sq.json, written by
    scripts/counter_summary.py from scripts/sq_counters.sh), dispatch-weighted over the kernel's instantiations."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', 'r*_bench_n%d_sq.json' % n)))
    if not files:
        return None
    with open(files[-1]) as fh:
        doc = json.load(fh)
    acc, wsum = {}, 0.0
    for name, e in doc.get('kernels', {}).items():
        if name.startswith(kernel_prefix):
            w = float(e.get('dispatches', 0))
            # weight by time: dispatches x resident cycles
            w *= float(e.get('derived', {}).get('kernel_cycles', 1.0))
            wsum += w
            for k, v in e.get('derived', {}).items():
                if k.endswith('_pct'):
                    acc[k] = acc.get(k, 0.0) + w * v
    if not wsum:
        return None
    out = {k: v / wsum for k, v in acc.items()}
    out['source'] = os.path.relpath(files[-1], ROOT)
    out['note'] = ('time-weighted over the kernel\'s launches; mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs), '
                   'wait_inst = issue stalls / wave cycles, wait_any = parked at s_waitcnt / barriers / sleeps')
    return out


# ------------------------------------------------------------------------------------------ CPU baselines
def blas_info():
    try:
        from threadpoolctl import threadpool_info
        return [{k: p.get(k) for k in ('user_api', 'internal_api', 'version', 'num_threads', 'threading_layer')}
                for p in threadpool_info()]
    except Exception:
        return None


def cpu_baseline(X, g, pts, threads, budget_s=20.0):
    """B-lean of BASELINE.md: the oracle's SciPy path on the host cores, bounded sample, with the seconds of each
    phase (K build = cdist + exp + S diagonal; dpotrf; dtrsv + reduction) timed on the last evaluation."""
    import scipy.linalg
    from threadpoolctl import threadpool_limits
    from oracle import sds_ref
    from oracle.kcgp_ref import covK as ref_covK
    n = X.shape[0]
    with threadpool_limits(limits=threads):
        t0 = time.perf_counter()
        vals, k = [], 0
        while True:
            vals.append(sds_ref.gp_log_marginal(X, g, pts[k % len(pts)]))
            k += 1
            el = time.perf_counter() - t0
            if el > budget_s or k >= 8:
                break
        # one more evaluation, phase by phase (the same calls gp_log_marginal makes)
        h = pts[0]
        p0 = time.perf_counter()
        K = ref_covK.RBF(np.log(h[0]), np.log(h[1])).getCovMatrix(x=X, mode='train')
        K[np.diag_indices(n)] += sds_ref.s_diag(np.diagonal(K).copy(), h[2])
        p1 = time.perf_counter()
        L = scipy.linalg.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
        p2 = time.perf_counter()
        sds_ref.log_marginal_lean(g, L)
        p3 = time.perf_counter()
        del K, L
    return dict(value=k / el, unit='evals/s', cores=int(threads), kind='port',
                sample='%d evaluations of the same N=%d workload (first hyper-parameter points), %.1f s' % (k, n, el),
                phase_seconds={'kbuild_cdist_exp': p1 - p0, 'dpotrf': p2 - p1, 'dtrsv_reduce': p3 - p2},
                dpotrf_gflops=n ** 3 / 3.0 / max(p2 - p1, 1e-9) / 1e9, blas=blas_info(),
                host_logical_cpus=os.cpu_count()), vals


def cpu_baseline_ref(X, g, pts, threads, cap_s=150.0):
    """B-ref of BASELINE.md: ONE proposal in the reference's literal op sequence (sliceSample.py:136-147,165-207:
    dense S, dpotrf, np.linalg.solve on a triangle, np.dot, inv(S), dpotrf, inv(K+S)) through the oracle restatement.
    ~11.3 N^3 flops: timed at N first on a 2048-point slice; if that predicts more than cap_s at full N the sample is
    the 4096-point slice instead, and says so."""
    from threadpoolctl import threadpool_limits
    from oracle import sds_ref
    from oracle.kcgp_ref import covK as ref_covK
    n = X.shape[0]

    def one(m):
        Xm, gm, h = X[:m], g[:m], pts[0]
        t0 = time.perf_counter()
        Kc = ref_covK.RBF(np.log(h[0]), np.log(h[1])).getCovMatrix(x=Xm, mode='train')
        _, K_S, m_, C_, L_ = sds_ref.aux_var_model_literal(np.zeros(m), Kc, h[2], gm)
        _ = np.dot(C_, np.zeros(m)) + m_
        _ = sds_ref.log_marginal_literal(gm, K_S, L_)
        return time.perf_counter() - t0

    with threadpool_limits(limits=threads):
        probe_n = min(n, 2048)
        t_probe = one(probe_n)
        predicted = t_probe * (n / probe_n) ** 3
        if n > probe_n and predicted <= cap_s:
            m, t = n, one(n)
        elif n > probe_n:
            m = min(n, 4096)
            t = one(m)
        else:
            m, t = probe_n, t_probe
    return dict(value=1.0 / t, unit='SDS proposals/s (reference op sequence, ~11.3 N^3 flops)', seconds_per_proposal=t,
                cores=int(threads), kind='port', n_sample=m,
                sample='one literal proposal at N=%d%s' % (m, '' if m == n else
                                                          ' (full N=%d predicted %.0f s from the N=%d probe: over the %.0f s cap)'
                                                          % (n, predicted, probe_n, cap_s)))


# ------------------------------------------------------------------------------------------ the work
class StubPlan(object):
    """GPMC_BENCH_STUB=1: no GPU, no library -- only so that the multi-rank plumbing of this script (spawn,
    rendez-vous, barriers, max-over-ranks, one JSON line) can be tested on CPU with gloo.  Never a measurement."""

    def __init__(self, n):
        self.n = n

    def eval_async(self, hyp, slot=0):
        time.sleep(1e-3)

    def result(self, slot=0):
        return (-1.0, 0.0, 0.0, 0)

    def drain(self):
        return {}

    def close(self):
        pass


def device_gp_draw(X, kernel, truth, seed):
    """g = chol(K(truth) + sn^2 I) z formed on the device with the library's own f64 kernels (bench INPUT for the extra
    configs: at N = 16384 the host Cholesky of synth_inputs would take most of a minute).  truth = (ell.., sf, sn)."""
    import torch
    from gpmc import covK, ops
    n = X.shape[0]
    dev = torch.device('cuda', torch.cuda.current_device())
    Xd = torch.as_tensor(X, dtype=torch.float64).to(dev)
    kobj, sn = covK.kernel_from_hyp(truth, kernel, torch.float64)
    kp = kobj.kparams()
    kp[1] = sn * sn
    np_ = ops.pad(n)
    W = torch.zeros((np_ + ops.PAD, np_), dtype=torch.float64, device=dev)
    ops.kbuild(kernel, Xd, torch.as_tensor(kp, dtype=torch.float64).to(dev), W, np_, np_, ops.KB_LOWER | ops.KB_PAD_IDENTITY)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    ops.potrf(W, np_ + ops.PAD, np_, info, ops.potrf_scratch(np_, dev))
    if int(info.item()) != 0:
        raise SystemExit('device_gp_draw: truth kernel matrix not positive definite (info=%d)' % int(info.item()))
    z = torch.as_tensor(np.random.RandomState(seed).standard_normal(n)).to(dev)
    g = torch.empty(n, dtype=torch.float64, device=dev)
    ops.trmv_lower(W, n, z, None, g)
    return g.cpu().numpy()


def time_config(name, n, d, kernel, dtype_name, steps, check_against_scipy, chains=0):
    """One of BASELINE.json's other configs at its per-GPU size: ms per log-marginal evaluation (sync-each, like the
    headline), TF/s of N^3/3 and the fraction of the dtype's matrix peak.  f32: the same evaluation in f64 beside it and
    the stated tolerance |f32 - f64| <= 2e-4 |f64| asserted (tests/test_gpu_kernels.py::test_config5_shape_f32_vs_f64)."""
    import torch
    from gpmc import core
    rng = np.random.RandomState(124)
    X = rng.uniform(0., 10., (n, d))
    if kernel == 'rbf':
        truth = np.array([1.0, 10.0, 1.2])
        pts = [np.array([1.3, 8.0, 1.0]), np.array([0.8, 12.0, 1.5]), np.array([2.0, 9.0, 0.9]), np.array([1.0, 10.0, 1.2])]
    else:
        ell = rng.uniform(0.5, 2.0, d)                                  # ell_k ~ U(0.5, 2)
        truth = np.concatenate([ell * 3.0, [10.0, 1.2]])
        pts = [np.concatenate([ell * s, [sf, sn]]) for s, sf, sn in ((3.0, 10.0, 1.2), (2.5, 8.0, 1.0), (3.5, 12.0, 1.5))]
    g = device_gp_draw(X, kernel, truth, 7)
    dt = torch.float32 if dtype_name == 'f32' else torch.float64
    plan = core.LogMarginalPlan(X, g, kernel, dtype=dt)
    vals = [plan(p) for p in pts]                                       # eager, eager, record (+ replay)
    plan(pts[0])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        plan.eval_async(pts[i % len(pts)], 0)
        r = plan.result(0)
        if r[3] != 0:
            raise SystemExit('%s: factorisation reported info=%d' % (name, r[3]))
    ms = 1e3 * (time.perf_counter() - t0) / steps
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    npipe = min(steps, RING)
    for i in range(npipe):                                              # back to back (never the quoted rate)
        plan.eval_async(pts[i % len(pts)], i)
    plan.drain()
    torch.cuda.synchronize()
    ms_pipe = 1e3 * (time.perf_counter() - t0) / npipe
    multi = None
    if chains > 1:
        # `chains` independent chains on this one card, each strictly sequential (its next evaluation is enqueued when
        # its previous result has been read), interleaved by this host thread: whole-card evaluations per second
        others = [core.LogMarginalPlan(X, g, kernel, dtype=dt) for _ in range(chains - 1)]
        for q in others:
            for p in pts:
                q(p)
        ps = [plan] + others
        for q in ps:
            q(pts[0])                                                    # (plans alive together re-record: core.py)
            q(pts[1])
            q(pts[2])
        torch.cuda.synchronize()
        for q in ps:
            q.eval_async(pts[0], 0)
        t0 = time.perf_counter()
        for i in range(steps):
            for q in ps:
                r = q.result(0)
                if r[3] != 0 or r[0] != vals[i % len(pts)]:
                    raise SystemExit('%s: a concurrent chain returned %r, alone %r' % (name, r, vals[i % len(pts)]))
                q.eval_async(pts[(i + 1) % len(pts)], 0)
        for q in ps:
            q.result(0)
        dtc = time.perf_counter() - t0
        multi = {'chains': chains, 'evals_per_s': steps * chains / dtc, 'ms_per_eval_per_chain': 1e3 * dtc / steps,
                 'note': 'independent chains on ONE card, one process, each sync-each; values checked bit for bit against the single chain'}
        for q in others:
            q.close()
    plan.close()
    peak = 157.3 if dtype_name == 'f32' else FP64_MATRIX_PEAK_TF
    tf = n ** 3 / 3.0 / (ms * 1e-3) / 1e12
    out = {'name': name, 'workload': '%s, N=%d, d=%d, %s' % ('RBF' if kernel == 'rbf' else 'ARD Matern-5/2', n, d, dtype_name),
           'ms_per_eval': ms, 'evals_per_s': 1e3 / ms, 'steps': steps, 'tflops': tf, 'peak_tflops': peak, 'frac_of_peak': tf / peak,
           'enqueue': 'sync-each (result read before the next evaluation)', 'ms_per_eval_pipelined': ms_pipe}
    if multi:
        out['concurrent_chains'] = multi
    if dtype_name == 'f32':
        p64 = core.LogMarginalPlan(X, g, kernel, dtype=torch.float64, settle=False)
        v64 = [p64(p) for p in pts]
        p64.close()
        rel = max(abs(a - b) / abs(b) for a, b in zip(vals, v64))
        if rel > 2e-4:
            raise SystemExit('%s: f32 log-marginal differs from f64 by %.3g relative (stated tolerance 2e-4)' % (name, rel))
        out['f32_vs_f64_rel_err'] = rel
        out['tolerance'] = 2e-4
    def check():          # run after every config has been timed: the host BLAS threads it wakes disturb the timings that follow
        from oracle import sds_ref                                       # checker only
        ref = sds_ref.gp_log_marginal(X, g, pts[0], kernel=('rbf' if kernel == 'rbf' else 'matern52ard'))
        rel = abs(vals[0] - ref) / abs(ref)
        if rel > 1e-8:
            raise SystemExit('%s: parity failure vs SciPy: %r vs %r' % (name, vals[0], ref))
        out['scipy_rel_err'] = rel
    return out, (check if check_against_scipy else None)


def timed_steps(plan, pts, steps, pipelined):
    """K steps; returns the last result.  sync-each: read step i before enqueuing i+1.  pipelined: back to back on the
    ring of result slots (a slot is read -- long complete -- before it is re-armed)."""
    last = None
    if pipelined:
        for i in range(steps):
            if i >= RING:
                last = plan.result(i % RING)
            plan.eval_async(pts[i % len(pts)], i % RING)
        for i in range(max(steps - RING, 0), steps):
            last = plan.result(i % RING)
    else:
        for i in range(steps):
            plan.eval_async(pts[i % len(pts)], 0)
            last = plan.result(0)
            if last[3] != 0:
                raise SystemExit('factorisation reported info=%d at step %d' % (last[3], i))
    return last


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--warmup', type=int, default=4)
    ap.add_argument('--n', type=int, default=8192)
    ap.add_argument('--d', type=int, default=2)
    ap.add_argument('--pipelined', action='store_true', help='time back-to-back enqueues instead of read-each-step')
    ap.add_argument('--no-cpu', action='store_true', help='skip both host baselines')
    ap.add_argument('--no-cpu-ref', action='store_true', help='skip the literal-sequence host baseline (B-ref)')
    ap.add_argument('--no-extras', action='store_true', help='skip the SDS-proposal and other-config timings of the default run')
    ap.add_argument('--metric-only', action='store_true',
                    help='launch nothing but the metric\'s kernels (for rocprofv3 runs): no host baselines, no extras')
    ap.add_argument('--full', action='store_true', help='= --sds --sds-cpu --chain 10')
    ap.add_argument('--sds', action='store_true', help='also time full SDS proposals (reported in "extra")')
    ap.add_argument('--sds-cpu', action='store_true', help='(kept for compatibility; B-ref is in the default line)')
    ap.add_argument('--chain', type=int, default=0, metavar='K',
                    help='also run K iterations of the real sampler (surrogate_slice_sampling) on these inputs: iterations/s')
    a = ap.parse_args()
    if a.full:
        a.sds, a.chain = True, (a.chain or 10)
    if a.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        return spawn_ranks(a.gpus, sys.argv[1:])

    import torch
    import torch.distributed as dist
    stub = os.environ.get('GPMC_BENCH_STUB') == '1'
    from gpmc import chains
    if not stub and not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (the gpmc product path has no CPU fallback)')
    rank, world = chains.init_distributed()          # binds cuda:LOCAL_RANK first, then joins the RCCL group
    if world != a.gpus:
        raise SystemExit('bench.py: --gpus %d but WORLD_SIZE=%d' % (a.gpus, world))
    dev = None if stub else torch.device('cuda', torch.cuda.current_device())
    threads = max(1, effective_cpus() // world)

    if stub:
        rng = np.random.RandomState(124 + rank)
        X, g, pts = rng.uniform(0, 10, (a.n, a.d)), rng.standard_normal(a.n), rng.uniform(0.5, 5., (16, 3))
        plan = StubPlan(a.n)
    else:
        import gpmc
        from gpmc import _lib, core
        for kv in filter(None, os.environ.get('GPMC_BENCH_OPTS', '').split(',')):     # development: library options for this run
            k, v = kv.split('=')
            gpmc.set_option(k, int(v))
        X, g, pts = synth_inputs(a.n, a.d, seed=124 + rank, threads=threads)
        plan = core.LogMarginalPlan(X, g, 'rbf')   # default path: the launch sequence recorded once, replayed per step

    grouped = dist.is_available() and dist.is_initialized()     # world > 1, or one rank with GPMC_FORCE_DIST=1 (RCCL rehearsal)
    nccl = grouped and dist.get_backend() == 'nccl'

    def barrier():
        if grouped:
            if nccl:
                dist.barrier(device_ids=[dev.index])
            else:
                dist.barrier()
        if dev is not None:
            torch.cuda.synchronize()

    timed_steps(plan, pts, a.warmup, a.pipelined)
    barrier()
    t0 = time.perf_counter()
    last = timed_steps(plan, pts, a.steps, a.pipelined)
    barrier()
    dt = time.perf_counter() - t0
    if last is not None and last[3] != 0:
        raise SystemExit('factorisation reported info=%d' % last[3])

    # spread: three more blocks of the same K steps (never `value`; the headline block is block 0)
    block_ms = [1e3 * dt / a.steps]
    for _ in range(3 if not stub else 0):
        if dev is not None:
            torch.cuda.synchronize()
        tb = time.perf_counter()
        timed_steps(plan, pts, a.steps, a.pipelined)
        if dev is not None:
            torch.cuda.synchronize()
        block_ms.append(1e3 * (time.perf_counter() - tb) / a.steps)

    # the other enqueue form, same K steps (reported beside the headline, never as `value`)
    timed_steps(plan, pts, min(a.warmup, 2), not a.pipelined)
    if dev is not None:
        torch.cuda.synchronize()
    t1 = time.perf_counter()
    timed_steps(plan, pts, a.steps, not a.pipelined)
    if dev is not None:
        torch.cuda.synchronize()
    dt_other = time.perf_counter() - t1

    k_ms = k_launches = k_flops = 0.0
    d_ms = d_launches = d_flops = 0.0
    dt_instr = 0.0
    if not stub:
        # roofline pass: the SAME K steps again with eager launches, every launch of the dominant kernel (trailing
        # SYRK update) bracketed by HIP events recorded on the stream it is launched on.  A graph replay exposes no
        # per-kernel boundaries to events, hence the second pass; its wall time is ms_per_step_instrumented.
        plan.close()
        gpmc.set_option('use_graph', 0)
        plan = core.LogMarginalPlan(X, g, 'rbf')
        timed_steps(plan, pts, min(a.warmup, 2), True)
        torch.cuda.synchronize()
        _lib.profile_begin()
        t1 = time.perf_counter()
        timed_steps(plan, pts, a.steps, True)
        torch.cuda.synchronize()
        dt_instr = time.perf_counter() - t1
        (k_ms, k_launches, k_flops), (d_ms, d_launches, d_flops) = _lib.profile_read()

    tmax = torch.tensor([dt], dtype=torch.float64, device=dev if (nccl or (world == 1 and dev is not None)) else 'cpu')
    if grouped:
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tmax = float(tmax.item())

    # host baselines (rank 0, N = 1 only) + parity of the timed path against the oracle on this rank's first points
    cpu = cpu_ref = None
    if rank == 0 and world == 1 and not stub and not a.no_cpu and not a.metric_only:
        cpu, vals = cpu_baseline(X, g, pts, threads)
        for j, v in enumerate(vals[:2]):
            got = plan(pts[j])
            if abs(got - v) > 1e-8 * abs(v):
                raise SystemExit('parity failure vs SciPy at point %d: %r vs %r' % (j, got, v))
        if not a.no_cpu_ref:
            cpu_ref = cpu_baseline_ref(X, g, pts, threads)

    if not stub:
        gpmc.set_option('use_graph', 1)             # (plans created from here on record and replay again)
    extra = {}
    if rank == 0 and not stub and not a.metric_only:
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
    default_extras = rank == 0 and world == 1 and not stub and not a.metric_only and not a.no_extras
    if (a.sds or default_extras) and rank == 0 and not stub:
        from gpmc import sds
        ws = sds.SDSWorkspace(X)
        f = torch.zeros(a.n, dtype=torch.float64, device=dev)
        gd = torch.as_tensor(g).to(dev)
        ws.aux_var_model(f, pts[0], gd, need_eta=True)       # the iteration's set-up (sliceSample.py:104-108)
        ws.begin(f, gd, ws.eta.clone())
        for j in range(3):                                   # eager, recording, first replay
            ws.proposal(pts[j % len(pts)])
        torch.cuda.synchronize()
        reps, per = len(pts), []
        for j in range(reps):                                # every hyper-parameter point once
            t1 = time.perf_counter()
            ws.proposal(pts[j])                              # G' and f' on the host: one sync on the happy path
            per.append((time.perf_counter() - t1, bool(ws._retried)))
        torch.cuda.synchronize()
        nretry = sum(1 for _, r in per if r)
        sds_dt = sum(t for t, _ in per) / reps
        happy = sorted(t for t, r in per if not r)
        extra['sds_proposals_per_s'] = 1.0 / sds_dt
        if happy:
            extra['sds_proposals_per_s_without_retry'] = 1.0 / happy[len(happy) // 2]
        extra['sds_flops_per_proposal'] = 4.0 * a.n ** 3 / 3.0     # inverse form (gpmc/sds.py)
        sds_tf = 4.0 * a.n ** 3 / 3.0 / sds_dt / 1e12
        extra['sds_roofline'] = {'bound': 'mfma', 'achieved': sds_tf, 'peak': FP64_MATRIX_PEAK_TF, 'unit': 'TFLOP/s',
                                 'frac': sds_tf / FP64_MATRIX_PEAK_TF, 'proposals': reps,
                                 'proposals_with_jitter_retry': nretry,
                                 'note': 'the reference\'s literal step (sliceSample.py:131-163): K build, chol(K+S) with g and '
                                         's I riding, R = sI - P P^T, m, chol(R + 1e-11 I), f\' = C eta + m, scalars read back; '
                                         'over all 16 hyper-parameter points, jitter retries of chol(R + 1e-11 I) included where a point needs them (each costs '
                                         'a rebuild of R and another factorisation); extra.sds_proposals_per_s_without_retry is the median '
                                         'proposal that needed none'}
        del ws
    if default_extras and a.n == 8192 and a.d == 2:
        # BASELINE.json's other configs at their per-GPU size, a few evaluations each (configs[1], configs[3] per GPU,
        # configs[4] per GPU)
        timed = [time_config('n2048', 2048, 2, 'rbf', 'f64', 40, True, chains=3),
                 time_config('n4096', 4096, 2, 'rbf', 'f64', 30, True, chains=3),
                 time_config('n16384_f32', 16384, 8, 'matern52ard', 'f32', 6, False)]
        for _, chk in timed:
            if chk:
                chk()
        extra['configs'] = {o['name']: o for o, _ in timed}
    if a.chain > 0 and rank == 0 and not stub:
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
        # two kernels carry the factorisation: the trailing-update GEMM (look-ahead phase) and the tile-DAG kernel (its
        # panels, and the tail = most of the flops at this size).  `roofline` describes the one with more kernel time in
        # the instrumented pass, `roofline_other` the other one.
        def roof(name, prefix, ms, launches, flops, note):
            ach = (flops / (ms * 1e-3)) / 1e12 if ms > 0 else 0.0
            tr, src = pmc_traffic(a.n, prefix) if (a.d == 2 and not stub) else (None, None)
            return {'bound': 'mfma', 'kernel': name, 'achieved': ach, 'peak': FP64_MATRIX_PEAK_TF, 'unit': 'TFLOP/s',
                    'frac': ach / FP64_MATRIX_PEAK_TF, 'launches_per_step': launches / max(a.steps, 1),
                    'avg_launch_us': 1e3 * ms / max(launches, 1), 'algorithmic_flops_per_launch': flops / max(launches, 1),
                    'kernel_ms_per_step': ms / max(a.steps, 1), 'note': note,
                    'traffic': tr, 'traffic_unit': 'HBM-side bytes per launch (PMC)', 'traffic_source': src,
                    'traffic_measured_in_run': False,
                    'counters': sq_counters(a.n, prefix) if (a.d == 2 and not stub) else None}
        r_gemm = roof('gpmc::gemm4_nt_kernel (trailing SYRK update of the blocked Cholesky)', 'gpmc::gemm4_nt_kernel',
                      k_ms, k_launches, k_flops, 'K = %d rank updates (outer block of the look-ahead at this size) on v_mfma_f64_4x4x4, 128x128 tiles' % (512 if a.n > 8064 else 384))
        r_dag = roof('gpmc::tail_kernel (tile-DAG factorisation: look-ahead panels and the last <= 6144 columns)',
                     'gpmc::tail_kernel', d_ms, d_launches, d_flops,
                     'one launch per panel / tail; bulk on v_mfma_f64_16x16x4 from 64x64 LDS tiles (operands prefetched '
                     'in the tail launch), critical path = the pivot chain (17 us per 64 columns) on '
                     'v_mfma_f64_4x4x4; panel launches overlap the GEMM on a second stream, so '
                     'kernel_ms_per_step of the two kernels add up to more than the step')
        roofline, roofline_other = (r_dag, r_gemm) if d_ms >= k_ms else (r_gemm, r_dag)
        for r in (roofline, roofline_other):
            r['whole_step_tflops'] = (a.n ** 3 / 3.0) / (tmax / a.steps) / 1e12
            r['measured_in'] = 'second pass over the same steps, eager launches, HIP events on the launch streams'
            r['ms_per_step_instrumented'] = 1e3 * dt_instr / a.steps
        other = 'pipelined' if not a.pipelined else 'sync_each'
        extra[other + '_evals_per_s'] = a.steps / dt_other          # this rank's own rate in the other enqueue form
        line = {
            'metric': 'GP log-marginal evals/sec (= slice-sample steps/sec) at N=%d' % a.n,
            'metric_note': 'one step = ONE log-marginal evaluation (K build + Cholesky + reduction, N^3/3 flops), the unit '
                           'SURVEY.md 8(d) defines; a full SDS proposal (the reference\'s literal step) is 4x: '
                           'extra.sds_proposals_per_s; a sampler iteration is several proposals: extra.chain with --full',
            'value': value, 'unit': 'evals/s', 'n_gpus': world, 'steps': a.steps, 'warmup': a.warmup,
            'ms_per_step': 1e3 * tmax / a.steps,
            'ms_per_step_blocks': {'blocks': len(block_ms), 'steps_each': a.steps, 'min': min(block_ms), 'max': max(block_ms),
                                   'median': float(np.median(block_ms)), 'all': block_ms,
                                   'note': 'this rank, block 0 = the headline block'},
            'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic' if not stub else 'STUB (no GPU work; plumbing test only)',
            'config': {'workload': 'RBF kernel, N=%d, d=%d, fp64, 1 chain per GPU (BASELINE.json configs[2])' % (a.n, a.d),
                       'global_batch': world, 'parallelism': 'chain-parallel x%d (no data-path collective)' % world,
                       'schedule': 'auto (look-ahead: one tile-DAG launch per %d-column panel beside the trailing GEMM while '
                                   '>6144 columns remain, then one tile-DAG launch for the rest)' % (512 if a.n > 8064 else 384),
                       'shared_card': 0 if stub else int(_lib.get_option('shared_card')),
                       'collective_backend': (dist.get_backend() if grouped else None),
                       'launch': 'recorded launch sequence (one fork/join hipGraph), replayed',
                       'enqueue': 'pipelined (back to back)' if a.pipelined else 'sync-each (result read before the next step)',
                       'g': 'GP draw chol(K+sn^2 I) z at the truth hyper-parameters (BASELINE.md 2)'},
            'roofline': roofline,
            'roofline_other': roofline_other,
            'cpu_baseline': cpu,
            'cpu_baseline_ref': cpu_ref,
        }
        if extra:
            line['extra'] = extra
        print(json.dumps(line), flush=True)
    plan.close()
    if grouped:
        barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
