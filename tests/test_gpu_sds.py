"""SDS proposal path, sampler mirror and chain drivers on the GPU vs the golden fixtures written by the
reference's own sampler (tests/golden, oracle/make_golden.py) and vs the CPU oracle.  -m gpu only."""
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import sds_ref                      # noqa: E402  (checker only)


@pytest.fixture(scope='module')
def gp():
    import gpmc
    from gpmc import core, covK, framework, ops, sds, sliceSample, tools
    ops.require_gpu()
    return dict(gpmc=gpmc, core=core, covK=covK, ops=ops, tools=tools, sds=sds, sliceSample=sliceSample,
                framework=framework)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_symv_trmv_syrk_units(gp):
    ops = gp['ops']
    rng = np.random.RandomState(0)
    for n in (1, 7, 64, 65, 200, 455, 1000):
        A = rng.standard_normal((n, n)); A = A + A.T
        x = rng.standard_normal(n); y0 = rng.standard_normal(n)
        np_ = ops.pad(n)
        W = torch.zeros((np_, np_), dtype=torch.float64, device='cuda')
        ops.pack_lower(torch.as_tensor(A).cuda(), W, np_, 0.0)
        xd, y0d = torch.as_tensor(x).cuda(), torch.as_tensor(y0).cuda()
        y = torch.empty(n, dtype=torch.float64, device='cuda')
        ops.symv_lower(W, n, xd, y)
        np.testing.assert_allclose(y.cpu().numpy(), A @ x, rtol=0, atol=1e-12 * (np.abs(A) @ np.abs(x)).max())
        ops.trmv_lower(W, n, xd, y0d, y)
        np.testing.assert_allclose(y.cpu().numpy(), np.tril(A) @ x + y0, rtol=0,
                                   atol=1e-12 * (np.abs(A) @ np.abs(x) + 1).max())
    n, kp = 300, 256
    P = rng.standard_normal((ops.pad(n), kp)); P[n:] = 0
    D = rng.standard_normal((n, n)); D = D + D.T
    np_ = ops.pad(n)
    Wd = torch.zeros((np_, np_), dtype=torch.float64, device='cuda')
    ops.pack_lower(torch.as_tensor(D).cuda(), Wd, np_, 0.0)
    ops.syrk_diff(Wd, torch.as_tensor(P).cuda(), kp, Wd, n, np_, 0.25)
    got = ops.unpack_lower(Wd, np_, n).cpu().numpy()
    ref = np.tril(D - P[:n] @ P[:n].T + 0.25 * np.eye(n))
    np.testing.assert_allclose(got, ref, rtol=0, atol=1e-11 * np.abs(ref).max())
    full = Wd.cpu().numpy()
    assert np.all(np.diag(full)[n:] == 1.0) and np.all(np.tril(full, -1)[n:, :] == 0.0)


@pytest.mark.parametrize('form', ['inv', 'syrk'])
@pytest.mark.parametrize('n', [8, 64, 200])
def test_aux_var_model_device_vs_reference_fixture(golden_dir, gp, n, form):
    """aux_var_model outputs of the reference (sliceSample.py:165-207) on the fixture's inputs, for the default
    inverse form (R = S - S (K+S)^-1 S, 4N^3/3 flops) and the first implementation (R = K - V^T V, 8N^3/3)."""
    z = _load(golden_dir, 'aux_n%d.npz' % n)
    ws = gp['sds'].SDSWorkspace(z['x'], form=form)
    f, g = torch.as_tensor(z['f']).cuda(), torch.as_tensor(z['g_in']).cuda()
    G = ws.aux_var_model(f, z['hyp'], g, need_eta=True)
    assert abs(G - float(z['curG'])) <= 1e-8 * abs(float(z['curG']))          # the north_star bar
    L = ws.L_KS().cpu().numpy()
    np.testing.assert_allclose(L, z['L'], rtol=0, atol=1e-11 * np.abs(z['L']).max())
    m = ws.m.cpu().numpy()
    np.testing.assert_allclose(m, z['m'], rtol=0, atol=1e-8 * np.abs(z['m']).max())
    C = ws.chol_R().cpu().numpy()
    # R = K - V^T V has rounding-level eigenvalues: compare C through C C^T (backward error), not entrywise
    np.testing.assert_allclose(C @ C.T, z['C'] @ z['C'].T, rtol=0, atol=1e-8)
    # eta = C^-1 (f - m) and f' = C eta + m must reproduce f for unchanged hyper-parameters
    fp = ws.propose_f(ws.eta.clone()).cpu().numpy()
    np.testing.assert_allclose(fp, z['f'], rtol=0, atol=1e-6 * max(1.0, np.abs(z['f']).max()))


def test_aux_var_model_protocol_function(golden_dir, gp):
    """the reference-shaped function aux_var_model(f, K, sn, g) on a dense K"""
    z = _load(golden_dir, 'aux_n64.npz')
    g, K_S, m, C, L = gp['sliceSample'].aux_var_model(z['f'], z['K'], float(z['hyp'][2]), g=z['g_in'])
    np.testing.assert_allclose(g.cpu().numpy(), z['g'], rtol=0, atol=0)
    np.testing.assert_allclose(K_S.cpu().numpy(), z['K_S'], rtol=1e-15)
    np.testing.assert_allclose(L.cpu().numpy(), z['L'], rtol=0, atol=1e-11 * np.abs(z['L']).max())
    np.testing.assert_allclose(m.cpu().numpy(), z['m'], rtol=0, atol=1e-8 * np.abs(z['m']).max())
    Cn = C.cpu().numpy()
    np.testing.assert_allclose(Cn @ Cn.T, z['C'] @ z['C'].T, rtol=0, atol=1e-8)


@pytest.mark.parametrize('n', [200, 1100])
def test_aux_var_model_protocol_function_general_K(gp, n):
    """a dense K with a NON-constant diagonal, so S of sliceSample.py:183-190 goes through the per-element expression
    (it equals sn^2 up to rounding whatever K_ii is); against the oracle's literal restatement (dense S, solve, inv)"""
    from oracle.kcgp_ref import covK as ocov
    rng = np.random.RandomState(n)
    x = rng.uniform(0, 10, (n, 2))
    d = rng.uniform(0.7, 1.3, n)
    K = ocov.RBF(np.log(1.5), np.log(3.0)).getCovMatrix(x=x, mode='train') * np.outer(d, d)
    f, g_in, sn = rng.standard_normal(n), rng.standard_normal(n) * 3., 0.9
    g0, KS0, m0, C0, L0 = sds_ref.aux_var_model_literal(f, K, sn, g=g_in)
    g, K_S, m, C, L = gp['sliceSample'].aux_var_model(f, K, sn, g=g_in)
    np.testing.assert_allclose(K_S.cpu().numpy(), KS0, rtol=1e-14)
    np.testing.assert_allclose(L.cpu().numpy(), L0, rtol=0, atol=1e-11 * np.abs(L0).max())
    np.testing.assert_allclose(m.cpu().numpy(), m0, rtol=0, atol=1e-8 * np.abs(m0).max())
    Cn = C.cpu().numpy()
    np.testing.assert_allclose(Cn @ Cn.T, C0 @ C0.T, rtol=0, atol=1e-8)


def test_log_gamma_matches_reference_fixture(golden_dir, gp):
    z = _load(golden_dir, 'log_gamma.npz')
    for i, row in enumerate(z['x']):
        a, b = gp['sliceSample'].log_gamma(row, z['k'], z['theta'], True)
        np.testing.assert_allclose(a, z['logG_inv'][i], rtol=1e-14, atol=1e-14)
        np.testing.assert_allclose(b, z['grad_inv'][i], rtol=1e-14, atol=1e-14)


@pytest.mark.parametrize('name', ['trace_framework_n64.npz', 'trace_demo_n48_d2.npz', 'trace_late_n32.npz',
                                  'trace_unfreeze_n8.npz'])
def test_seeded_chain_follows_reference_trace(golden_dir, gp, name):
    """Chains run by the reference sampler (fixtures) vs the device sampler on the same seed: identical
    accept/reject decisions (proposal counts, RNG stream position) and hyper-parameter history.  Latent f is
    compared loosely on purpose -- see tests/test_oracle_golden.py::test_seeded_trace_lean_follows_reference."""
    z = _load(golden_dir, name)
    np.random.seed(int(z['seed']))
    histF, histHyp, ntries = gp['framework'].run_chain(z['x'], z['y'], int(z['iters']), z['hyp0'],
                                                       iter0=int(z['iter0']))
    tail = np.random.uniform(size=4)
    np.testing.assert_array_equal(ntries, z['ntries'])
    np.testing.assert_allclose(histHyp, z['histHyp'], rtol=1e-12)
    np.testing.assert_array_equal(tail, z['rng_tail'])
    np.testing.assert_allclose(histF, z['histF'], rtol=0, atol=5e-2)


def test_chain_recovers_hyperparameters_statistically(gp):
    """posterior of (ll, sf, sn) on synthetic data with known truth brackets the truth"""
    rng = np.random.RandomState(7)
    n = 300
    x = np.sort(rng.uniform(0, 30, (n, 1)), axis=0)
    truth = (2.0, 8.0, 1.0)
    K = sds_ref.covK.RBF(np.log(truth[0]), np.log(truth[1])).getCovMatrix(x=x, mode='train')
    y = np.clip(np.linalg.cholesky(K + truth[2] ** 2 * np.eye(n)) @ rng.standard_normal(n) + 50., 1., 99.)
    np.random.seed(3)
    histF, histHyp, ntries = gp['framework'].run_chain(x, y, 600, gp['framework'].FRAMEWORK_HYP0)
    post = histHyp[:, 150:]
    lo, hi = np.percentile(post, 1, axis=1), np.percentile(post, 99, axis=1)
    assert lo[0] < truth[0] < hi[0] and lo[1] < truth[1] < hi[1]
    # sn is frozen at its initial value for the first 500 iterations (sliceSample.py:133) and sampled afterwards
    assert np.all(histHyp[2, :500] == gp['framework'].FRAMEWORK_HYP0[2]) and np.all(histHyp[2, 500:] > 0)
    assert np.ptp(histHyp[2, 500:]) > 0
    assert np.all(np.isfinite(histF))


def test_framework_and_demo_drivers(gp):
    fw = gp['framework']
    x, y = fw.synthetic_demo_data(200, 124)
    xo, yo = sds_ref.synthetic_demo_data(200, 124)
    np.testing.assert_array_equal(x, xo)
    np.testing.assert_allclose(y, yo, rtol=0, atol=1e-9 * np.abs(yo).max())
    y = np.clip(y, 1., 99.)
    np.random.seed(1)
    histHyp, histF = fw.mcmcUpdate({'x': x, 'y': y}, 3)
    assert histHyp.shape == (3, 3) and histF.shape == (200, 3)
    np.random.seed(1)
    F = fw.Framework(np.hstack([y.reshape(-1, 1), x]))
    hf, hh = F.runSimulMCMC(2)
    assert hf.shape == (200, 2) and hh.shape == (3, 2)


def test_kcGP_kcMCMC_aliases(gp):
    import kcGP
    import kcMCMC
    assert kcGP.covK.RBF is gp['covK'].RBF and kcMCMC.sdsK is kcMCMC.sliceSample
    L = kcGP.tools.jitchol(np.eye(3) * 4.0).cpu().numpy()
    np.testing.assert_allclose(L, 2 * np.eye(3))


# ------------------------------------------------------------------ BASELINE.json's full sizes
def _bench_like(n, d, seed=124):
    rng = np.random.RandomState(seed)
    return rng.uniform(0., 10., (n, d)), rng.standard_normal(n) * 5.0


@pytest.mark.parametrize('n', [4096, 8192])
def test_full_size_log_marginal(gp, n):
    """configs 3 and 4 (N = 8192 / 4096, d = 2, fp64): against SciPy directly (a few seconds of CPU), plus
    size-independent properties: bitwise repeatability and the residual (L L^T) v = (K + S) v."""
    X, g = _bench_like(n, 2)
    hyp = [1.0, 10.0, 1.2]
    plan = gp['core'].LogMarginalPlan(X, g, 'rbf')
    a = plan(hyp)
    b = plan([2.0, 7.0, 0.9])
    a2 = plan(hyp)
    assert a == a2 and a != b
    ref = sds_ref.gp_log_marginal(X, g, hyp)
    assert abs(a - ref) <= 1e-8 * abs(ref), (a, ref)
    W, mp, np_, ld = plan.factor_view()
    ops = gp['ops']
    v = torch.as_tensor(np.random.RandomState(1).standard_normal(n)).cuda()
    Lt_v = torch.mv(torch.tril(W[:n, :n]).T, v)                 # checker arithmetic (torch), not the product path
    LLt_v = torch.empty(n, dtype=torch.float64, device='cuda')
    ops.trmv_lower(W, n, Lt_v, None, LLt_v)
    kobj, sn = gp['covK'].kernel_from_hyp(hyp)
    KS = kobj.getCovMatrix(x=X, mode='train', diag_add=gp['covK'].s_diag_scalar(kobj.kparams()[0], sn))
    ref_v = torch.mv(KS, v)
    rel = float((LLt_v - ref_v).abs().max() / ref_v.abs().max())
    assert rel < 1e-11, rel
    plan.close()


def test_full_size_sds_proposal_properties(gp):
    """one full SDS proposal at N = 4096: f' = C C^-1 (f - m) + m returns f; alpha^T alpha matches the
    quadratic form through an independent route (g^T (K+S)^-1 g via two triangular products)."""
    n = 4096
    X, g0 = _bench_like(n, 2)
    ws = gp['sds'].SDSWorkspace(X)
    f = torch.as_tensor(np.random.RandomState(2).standard_normal(n)).cuda()
    g = torch.as_tensor(g0).cuda()
    G = ws.aux_var_model(f, [1.0, 10.0, 1.2], g, need_eta=True)
    assert np.isfinite(G)
    fp = ws.propose_f(ws.eta.clone())
    assert float((fp - f).abs().max()) < 1e-5
    ref = sds_ref.gp_log_marginal(X, g0, [1.0, 10.0, 1.2])
    assert abs(G - ref) <= 1e-8 * abs(ref)


def test_sds_forms_agree_at_size(gp):
    """N = 4096 (big tiles, trapezoidal riding rows, triangular SYRK): the two algebraic forms of the SDS block give the
    same L bit for bit (the riding rows do not feed back), the same log-marginal, m to 1e-9 and R + 1e-11 I to 1e-8 in
    C C^T; each reproduces f through eta."""
    n = 4096
    X, g0 = _bench_like(n, 2)
    f = torch.as_tensor(np.random.RandomState(2).standard_normal(n)).cuda()
    g = torch.as_tensor(g0).cuda()
    out = {}
    for form in ('inv', 'syrk'):
        ws = gp['sds'].SDSWorkspace(X, form=form)
        G = ws.aux_var_model(f, [1.3, 9.0, 1.1], g, need_eta=True)
        C = ws.chol_R()
        fp = ws.propose_f(ws.eta.clone())
        assert float((fp - f).abs().max()) < 1e-5
        out[form] = (G, ws.L_KS().clone(), ws.m.clone(), (C @ C.t()))
        del ws
    assert out['inv'][0] == out['syrk'][0]
    assert torch.equal(out['inv'][1], out['syrk'][1])
    mi, ms = out['inv'][2], out['syrk'][2]
    assert float((mi - ms).abs().max()) <= 1e-9 * float(ms.abs().max())
    assert float((out['inv'][3] - out['syrk'][3]).abs().max()) <= 1e-8


def test_config5_shape_f32_vs_f64(gp):
    """config 5 at its own size (ARD Matern-5/2, N = 16384, d = 8): the fp64 device path against SciPy (1e-8 relative,
    the north_star bar), and the fp32 factorisation (float storage, look-ahead schedule) against the fp64 one with the
    stated fp32 tolerance of 2e-4 relative on the log-marginal."""
    n, d = 16384, 8
    X, g = _bench_like(n, d, 5)
    ell = np.random.RandomState(2).uniform(0.5, 2.0, d) * 3.0
    hyp = np.concatenate([ell, [2.0, 0.7]])
    a64 = gp['core'].log_marginal(X, g, hyp, kernel='matern52ard', dtype=torch.float64)
    a32 = gp['core'].log_marginal(X, g, hyp, kernel='matern52ard', dtype=torch.float32)
    assert abs(a32 - a64) <= 2e-4 * abs(a64), (a32, a64)
    ref = sds_ref.gp_log_marginal(X, g, hyp, kernel='matern52ard')
    assert abs(a64 - ref) <= 1e-8 * abs(ref), (a64, ref)