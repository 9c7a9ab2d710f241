#!/usr/bin/env python3
"""Generate ``tests/golden/*.npz`` by running the REFERENCE's own sampler.

TEST INFRASTRUCTURE ONLY.  Runs in the build container only (it reads
``/root/reference``, which does not exist on the GPU box).  The reference's
``kcMCMC/sliceSample.py`` is loaded unmodified *by file path*; the git-ignored
``kcGP`` package it imports is supplied by ``oracle/kcgp_ref`` through
``sys.modules``.  Outputs are data only: inputs and the
reference's outputs on them.

    python oracle/make_golden.py [--ref /root/reference] [--out tests/golden]
"""
import argparse
import importlib.util
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import kcgp_ref, sds_ref  # noqa: E402


def load_reference(ref_root):
    sys.modules['kcGP'] = kcgp_ref
    sys.modules['kcGP.covK'] = kcgp_ref.covK
    sys.modules['kcGP.likK'] = kcgp_ref.likK
    sys.modules['kcGP.tools'] = kcgp_ref.tools
    path = os.path.join(ref_root, 'kcMCMC', 'sliceSample.py')
    spec = importlib.util.spec_from_file_location('ref_sliceSample', path)
    mod = importlib.util.module_from_spec(spec)
    sys.dont_write_bytecode = True
    spec.loader.exec_module(mod)
    return mod


def make_xy(n, d, seed, clip=True):
    """GP draw with the framework's initial hyper-parameters, shifted to the 0..100 score range."""
    rng = np.random.RandomState(seed)
    x = np.sort(rng.uniform(0., 10., (n, d)), axis=0) if d == 1 else rng.uniform(0., 10., (n, d))
    K = kcgp_ref.covK.RBF(np.log(1.0), np.log(10.0)).getCovMatrix(x=x, mode='train')
    L = np.linalg.cholesky(K + 1.2 ** 2 * np.eye(n))
    y = L @ rng.standard_normal(n) + 50.
    if clip:
        y = np.clip(y, 1., 99.)
    return x, y


def golden_aux(ref, out):
    for n, d, seed in ((8, 1, 1), (64, 2, 2), (200, 1, 3)):
        x, y = make_xy(n, d, seed)
        rng = np.random.RandomState(100 + seed)
        hyp = np.array([0.8 + 0.4 * rng.rand(), 6. + 6. * rng.rand(), 0.8 + 0.8 * rng.rand()])
        f = (y - y.mean()) * 0.5
        g_in = f + hyp[2] * rng.standard_normal(n)
        K = kcgp_ref.covK.RBF(np.log(hyp[0]), np.log(hyp[1])).getCovMatrix(x=x, mode='train')
        g, K_S, m, C, L = ref.aux_var_model(f, K, hyp[2], g=g_in)
        curG = -(np.dot(np.dot(g.T, np.linalg.inv(K_S)), g) / 2. + np.log(np.diag(L.T)).sum()
                 + g.shape[0] * np.log(2 * np.pi) / 2.)          # expression of sliceSample.py:122
        ita = np.linalg.solve(C, f - m)                          # sliceSample.py:108
        prop_f = np.dot(C, ita) + m                              # sliceSample.py:140
        np.savez_compressed(os.path.join(out, 'aux_n%d.npz' % n), x=x, y=y, f=f, hyp=hyp, g_in=g_in,
                            K=K, g=g, K_S=K_S, m=m, C=C, L=L, curG=curG, ita=ita, prop_f=prop_f)
        # g drawn by the reference itself (multivariate_normal on the legacy stream)
        np.random.seed(7 + n)
        g_drawn = ref.aux_var_model(f, K, hyp[2])[0]
        np.savez_compressed(os.path.join(out, 'gdraw_n%d.npz' % n), f=f, K=K, sn=hyp[2], seed=7 + n, g=g_drawn)


def golden_log_gamma(ref, out):
    k = np.asarray([1., 3., 3.])
    theta = np.asarray([1., 1.5, 3.])
    rng = np.random.RandomState(5)
    xs = np.vstack([rng.uniform(0.05, 12., (40, 3)), [[1., 10., 1.2]], [[0.35, 2.0, 0.2]]])
    lg_inv, gr_inv, lg_pl, gr_pl = [], [], [], []
    for row in xs:
        a, b = ref.log_gamma(row.copy(), k, theta, True)
        lg_inv.append(a), gr_inv.append(b)
        a, b = ref.log_gamma(row.copy(), k, theta, False)
        lg_pl.append(a), gr_pl.append(b)
    np.savez_compressed(os.path.join(out, 'log_gamma.npz'), x=xs, k=k, theta=theta,
                        logG_inv=np.asarray(lg_inv), grad_inv=np.asarray(gr_inv),
                        logG=np.asarray(lg_pl), grad=np.asarray(gr_pl))


_REAL_RBF = kcgp_ref.covK.RBF


class _CountingRBF(_REAL_RBF):
    count = 0

    def __init__(self, *a, **kw):
        _CountingRBF.count += 1
        _REAL_RBF.__init__(self, *a, **kw)


def golden_trace(ref, out, name, n, d, iters, hyp0, seed, iter0=0):
    """Seeded chain in the shape of framework.py:59-77 / demoRegression.py:15-32, driven by the reference sampler."""
    x, y = make_xy(n, d, 40 + n)
    kcgp_ref.covK.RBF = _CountingRBF
    try:
        np.random.seed(seed)
        hyp = np.asarray(hyp0, dtype=np.float64)
        f = np.zeros_like(y)
        histF = np.zeros((n, iters))
        histHyp = np.zeros((3, iters))
        ntries = np.zeros(iters, dtype=np.int64)
        for i in range(iters):
            _CountingRBF.count = 0
            f, hyp = ref.surrogate_slice_sampling(f, x, y, hyp, scale=np.asarray([10., 10., 5.]), iter=iter0 + i)
            histF[:, i] = f
            histHyp[:, i] = hyp
            ntries[i] = _CountingRBF.count - 1
        rng_tail = np.random.uniform(size=4)          # stream position check
    finally:
        kcgp_ref.covK.RBF = _REAL_RBF
    np.savez_compressed(os.path.join(out, name), x=x, y=y, hyp0=np.asarray(hyp0), seed=seed, iters=iters,
                        iter0=iter0, histF=histF, histHyp=histHyp, ntries=ntries, rng_tail=rng_tail)
    print('%s: n=%d iters=%d proposals=%d' % (name, n, iters, ntries.sum()))


def golden_trace_demo(ref, out, name, n, iters, seed):
    """Seeded chain at the reference's own demo sizes: data of the demoRegression.py:118-131 generator (seed 124, ll=5,
    sf=20, sn=2.5, x = arange(n); n = 455 there, BASELINE config 1 says ~200) clipped to the 1..99 score range, the
    demo's initial point (demoRegression.py:15), driven by the REFERENCE sampler.  Also stored, clearly marked as the
    ORACLE's (sds_ref literal restatement, same seed -- it reproduces the reference's decisions exactly, which this
    function asserts): the margin proposal - threshold of every try, i.e. how close each accept/shrink decision was."""
    x, y = sds_ref.synthetic_demo_data(n, 124)
    y = np.clip(y, 1., 99.)
    kcgp_ref.covK.RBF = _CountingRBF
    try:
        np.random.seed(seed)
        hyp = np.asarray(sds_ref.DEMO_HYP0, dtype=np.float64)
        f = np.zeros_like(y)
        histF = np.zeros((n, iters))
        histHyp = np.zeros((3, iters))
        ntries = np.zeros(iters, dtype=np.int64)
        for i in range(iters):
            _CountingRBF.count = 0
            f, hyp = ref.surrogate_slice_sampling(f, x, y, hyp, scale=np.asarray([10., 10., 5.]), iter=i)
            histF[:, i] = f
            histHyp[:, i] = hyp
            ntries[i] = _CountingRBF.count - 1
        rng_tail = np.random.uniform(size=4)
    finally:
        kcgp_ref.covK.RBF = _REAL_RBF
    np.random.seed(seed)
    marg = []
    oF, oH, oN = sds_ref.run_chain(x, y, iters, sds_ref.DEMO_HYP0, literal=True, margins=marg)
    assert np.array_equal(oN, ntries) and np.array_equal(oH, histHyp), 'oracle restatement left the reference trace'
    width = int(ntries.max())
    margins = np.full((iters, width), np.nan)
    for i, row in enumerate(marg):
        margins[i, :len(row)] = row
    np.savez_compressed(os.path.join(out, name), x=x, y=y, hyp0=np.asarray(sds_ref.DEMO_HYP0), seed=seed, iters=iters,
                        iter0=0, histF=histF, histHyp=histHyp, ntries=ntries, rng_tail=rng_tail,
                        oracle_margins=margins)
    print('%s: n=%d iters=%d proposals=%d  smallest |margin| %.3e' % (name, n, iters, ntries.sum(), np.nanmin(np.abs(margins))))


def golden_inf_mcmc(ref, out):
    """the reference's own inf_mcmc (sliceSample.py:234-284) on a small model object"""
    x, y = make_xy(60, 1, 77)
    xs = np.linspace(-1., 11., 23).reshape(-1, 1)
    rng = np.random.RandomState(8)
    f = (y - y.mean())[:, None] + rng.standard_normal((60, 5))
    hyp = np.array([1.1, 9.0, 1.3])
    model = sds_ref.Model(x, y, xs, kcgp_ref.covK.RBF(np.log(hyp[0]), np.log(hyp[1])),
                          kcgp_ref.likK.TruncatedGauss2(upper=100. - y.mean(), lower=0. - y.mean(),
                                                        log_sigma=np.log(hyp[2])))
    ym, lw, up, Fs2 = ref.inf_mcmc(f, model)
    np.savez_compressed(os.path.join(out, 'inf_mcmc_n60.npz'), x=x, y=y, xs=xs, f=f, hyp=hyp, ym=ym, ys_lw=lw,
                        ys_up=up, Fs2=Fs2)


def golden_elliptical(ref, out):
    """the reference's own elliptical_slice (sliceSample.py:15-74), seeded; nu recorded by re-deriving it"""
    x, y = make_xy(40, 1, 55)
    hyp = np.array([1.0, 10.0, 1.2])
    f0 = (y - y.mean()) * 0.3
    np.random.seed(31)
    K = kcgp_ref.covK.RBF(np.log(hyp[0]), np.log(hyp[1])).getCovMatrix(x=x, mode='train')
    nu = np.random.multivariate_normal(np.zeros_like(f0), K, 1).T.reshape((40,))      # what the reference draws first
    np.random.seed(31)
    f1 = ref.elliptical_slice(f0.copy(), x, y, hyp)
    tail = np.random.uniform(size=3)
    np.savez_compressed(os.path.join(out, 'elliptical_n40.npz'), x=x, y=y, hyp=hyp, f0=f0, nu=nu, f1=f1, seed=31,
                        rng_tail=tail)


def golden_synthetic(out):
    """demoRegression.py:116-131 is Python-2-only __main__ code and cannot be executed;
    this fixture is the oracle's restatement of it (documented as such in the test)."""
    x, y = sds_ref.synthetic_demo_data(455, 124)
    np.savez_compressed(os.path.join(out, 'synthetic_seed124.npz'), x=x, y=y)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--ref', default='/root/reference')
    ap.add_argument('--out', default=os.path.join(os.path.dirname(HERE), 'tests', 'golden'))
    a = ap.parse_args()
    os.makedirs(a.out, exist_ok=True)
    ref = load_reference(a.ref)
    golden_aux(ref, a.out)
    golden_log_gamma(ref, a.out)
    golden_trace(ref, a.out, 'trace_framework_n64.npz', 64, 1, 12, sds_ref.FRAMEWORK_HYP0, 2024)
    golden_trace(ref, a.out, 'trace_demo_n48_d2.npz', 48, 2, 10, sds_ref.DEMO_HYP0, 11)
    golden_trace(ref, a.out, 'trace_unfreeze_n8.npz', 8, 1, 520, sds_ref.FRAMEWORK_HYP0, 99)
    golden_trace(ref, a.out, 'trace_late_n32.npz', 32, 1, 8, sds_ref.FRAMEWORK_HYP0, 5, iter0=498)
    golden_trace_demo(ref, a.out, 'trace_demo_n200.npz', 200, 30, 5)
    golden_trace_demo(ref, a.out, 'trace_demo_n455.npz', 455, 20, 5)
    golden_inf_mcmc(ref, a.out)
    golden_elliptical(ref, a.out)
    golden_synthetic(a.out)


if __name__ == '__main__':
    main()
