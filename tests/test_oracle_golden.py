"""The CPU oracle against fixtures written by the reference's own sliceSample.py
(oracle/make_golden.py).  Pins oracle/sds_ref.py; kcGP itself stays parity-unpinned."""
import os

import numpy as np
import pytest

from oracle import sds_ref
from oracle.kcgp_ref import covK, tools

pytestmark = pytest.mark.filterwarnings('ignore')


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


@pytest.mark.parametrize('n', [8, 64, 200])
def test_aux_var_model_literal_matches_reference(golden_dir, n):
    z = _load(golden_dir, 'aux_n%d.npz' % n)
    g, K_S, m, C, L = sds_ref.aux_var_model_literal(z['f'], z['K'], float(z['hyp'][2]), g=z['g_in'])
    np.testing.assert_array_equal(g, z['g'])
    np.testing.assert_array_equal(K_S, z['K_S'])
    np.testing.assert_array_equal(L, z['L'])
    np.testing.assert_allclose(m, z['m'], rtol=0, atol=1e-12 * np.abs(z['m']).max())
    np.testing.assert_allclose(C, z['C'], rtol=0, atol=1e-12)
    assert sds_ref.log_marginal_literal(g, K_S, L) == pytest.approx(float(z['curG']), rel=1e-14)


@pytest.mark.parametrize('n', [8, 64, 200])
def test_aux_var_model_lean_within_tolerance(golden_dir, n):
    """The triangular form the HIP path computes vs the reference's solve/inv form.
    R = K - V^T V is ill-conditioned, so C is compared through C C^T and prop_f."""
    z = _load(golden_dir, 'aux_n%d.npz' % n)
    sn = float(z['hyp'][2])
    g, Sii, m, C, L = sds_ref.aux_var_model_lean(z['f'], z['K'], sn, g=z['g_in'])
    np.testing.assert_allclose(Sii, sn ** 2, rtol=1e-13)
    np.testing.assert_allclose(L, z['L'], rtol=0, atol=1e-12 * np.abs(z['L']).max())
    np.testing.assert_allclose(m, z['m'], rtol=0, atol=1e-9 * np.abs(z['m']).max())
    np.testing.assert_allclose(C @ C.T, z['C'] @ z['C'].T, rtol=0, atol=1e-9)
    assert sds_ref.log_marginal_lean(g, L) == pytest.approx(float(z['curG']), rel=1e-11)


@pytest.mark.parametrize('n', [8, 64, 200])
def test_g_draw_is_stream_compatible(golden_dir, n):
    """multivariate_normal(f, s*I) == f + sqrt(s)*standard_normal(N) on the legacy stream."""
    z = _load(golden_dir, 'gdraw_n%d.npz' % n)
    np.random.seed(int(z['seed']))
    Sii = sds_ref.s_diag(np.diagonal(z['K']), float(z['sn']))
    g = sds_ref.draw_g(z['f'], Sii)
    np.testing.assert_allclose(g, z['g'], rtol=0, atol=1e-13 * np.abs(z['g']).max())


def test_log_gamma(golden_dir):
    z = _load(golden_dir, 'log_gamma.npz')
    for i, row in enumerate(z['x']):
        a, b = sds_ref.log_gamma(row, z['k'], z['theta'], True)
        np.testing.assert_allclose(a, z['logG_inv'][i], rtol=1e-14, atol=1e-14)
        np.testing.assert_allclose(b, z['grad_inv'][i], rtol=1e-14, atol=1e-14)
        a, b = sds_ref.log_gamma(row, z['k'], z['theta'], False)
        np.testing.assert_allclose(a, z['logG'][i], rtol=1e-14, atol=1e-14)
        np.testing.assert_allclose(b, z['grad'][i], rtol=1e-14, atol=1e-14)


@pytest.mark.parametrize('name', ['trace_framework_n64.npz', 'trace_demo_n48_d2.npz',
                                  'trace_unfreeze_n8.npz', 'trace_late_n32.npz'])
def test_seeded_trace_literal(golden_dir, name):
    """Draw-for-draw reproduction of chains run by the reference sampler, including the
    per-iteration proposal count and the RNG stream position afterwards."""
    z = _load(golden_dir, name)
    np.random.seed(int(z['seed']))
    histF, histHyp, ntries = sds_ref.run_chain(z['x'], z['y'], int(z['iters']), z['hyp0'],
                                               literal=True, iter0=int(z['iter0']))
    tail = np.random.uniform(size=4)
    np.testing.assert_array_equal(ntries, z['ntries'])
    np.testing.assert_allclose(histHyp, z['histHyp'], rtol=1e-13)
    np.testing.assert_allclose(histF, z['histF'], rtol=0, atol=1e-7)
    np.testing.assert_array_equal(tail, z['rng_tail'])


@pytest.mark.parametrize('name', ['trace_demo_n200.npz', 'trace_demo_n455.npz'])
def test_seeded_trace_literal_at_the_reference_sizes(golden_dir, name):
    """the reference's own demo sizes (demoRegression.py:118-131 generator; 30 / 20 iterations): the literal
    restatement reproduces the reference run draw for draw, and the decision margins stored with the fixture"""
    z = _load(golden_dir, name)
    np.random.seed(int(z['seed']))
    marg = []
    histF, histHyp, ntries = sds_ref.run_chain(z['x'], z['y'], int(z['iters']), z['hyp0'], literal=True, margins=marg)
    tail = np.random.uniform(size=4)
    np.testing.assert_array_equal(ntries, z['ntries'])
    np.testing.assert_allclose(histHyp, z['histHyp'], rtol=1e-13)
    np.testing.assert_array_equal(tail, z['rng_tail'])
    for i, row in enumerate(marg):
        np.testing.assert_allclose(row, z['oracle_margins'][i][:len(row)], rtol=1e-9, atol=1e-9)


def test_seeded_trace_lean_follows_reference(golden_dir):
    """The lean (triangular) form takes the same accept/reject decisions on a short trace.

    Latent f is compared loosely ON PURPOSE: R = K - V^T V has eigenvalues at rounding level
    (hence the reference's 1e-11 jitter, sliceSample.py:205), so eta = C^-1 (f - m) amplifies
    last-bit differences of R by ~1/sqrt(1e-11); any re-ordering of the dense arithmetic moves
    f at the 1e-2 level while hyper-parameters and decisions stay put."""
    z = _load(golden_dir, 'trace_framework_n64.npz')
    np.random.seed(int(z['seed']))
    histF, histHyp, ntries = sds_ref.run_chain(z['x'], z['y'], int(z['iters']), z['hyp0'], literal=False)
    np.testing.assert_array_equal(ntries, z['ntries'])
    np.testing.assert_allclose(histHyp, z['histHyp'], rtol=1e-12)
    np.testing.assert_allclose(histF, z['histF'], rtol=0, atol=5e-2)


def test_synthetic_generator_fixture(golden_dir):
    """demoRegression.py:116-131 restated (the Py2 __main__ block itself cannot run): the fixture
    is the oracle's own output, kept to detect drift; x = arange(455), seed 124."""
    z = _load(golden_dir, 'synthetic_seed124.npz')
    x, y = sds_ref.synthetic_demo_data(455, 124)
    np.testing.assert_array_equal(x, z['x'])
    np.testing.assert_allclose(y, z['y'], rtol=1e-12)
    assert x.shape == (455, 1) and abs(y.mean() - 91.15) < 15


def test_kcgp_protocol_shapes():
    x = np.random.RandomState(0).uniform(0, 10, (20, 2))
    xs = np.random.RandomState(1).uniform(0, 10, (7, 2))
    k = covK.RBF(np.log(1.3), np.log(2.0))
    K = k.getCovMatrix(x=x, mode='train')
    assert K.shape == (20, 20) and np.allclose(np.diag(K), 4.0)
    assert k.getCovMatrix(x=x, z=xs, mode='cross').shape == (20, 7)
    assert k.getCovMatrix(z=xs, mode='self_test').shape == (7, 1)
    L = tools.jitchol(K + 0.1 * np.eye(20))
    assert np.allclose(np.triu(L, 1), 0) and np.allclose(L @ L.T, K + 0.1 * np.eye(20))
    B = np.random.RandomState(2).randn(20, 3)
    X = tools.solve_chol(L.T, B)
    assert np.allclose((K + 0.1 * np.eye(20)) @ X, B)
    with pytest.raises(np.linalg.LinAlgError):
        tools.jitchol(-np.eye(3))


def test_jitchol_jitter_retry():
    a = np.ones((4, 4))                     # rank one, diag > 0: dpotrf fails, jitter path succeeds
    L = tools.jitchol(a)
    assert np.allclose(L @ L.T, a, atol=1e-6)


def test_matern_ard_limits():
    x = np.random.RandomState(3).uniform(0, 3, (15, 4))
    k = covK.MaternARD52(np.log([1., 2., .5, 3.]), np.log(1.5))
    K = k.getCovMatrix(x=x, mode='train')
    assert np.allclose(np.diag(K), 2.25) and np.all(np.linalg.eigvalsh(K) > 0)
    assert np.allclose(K, K.T)


def test_inf_mcmc_restatement_matches_reference(golden_dir):
    """sliceSample.py:234-284 run by the reference itself (fixture) vs the oracle's restatement"""
    from oracle.kcgp_ref import likK
    z = _load(golden_dir, 'inf_mcmc_n60.npz')
    hyp = z['hyp']
    model = sds_ref.Model(z['x'], z['y'], z['xs'], covK.RBF(np.log(hyp[0]), np.log(hyp[1])),
                          likK.TruncatedGauss2(upper=100. - z['y'].mean(), lower=0. - z['y'].mean(),
                                               log_sigma=np.log(hyp[2])))
    ym, lw, up, Fs2 = sds_ref.inf_mcmc(z['f'], model)
    np.testing.assert_allclose(ym, z['ym'], rtol=1e-10)
    np.testing.assert_allclose(lw, z['ys_lw'], rtol=1e-9)
    np.testing.assert_allclose(up, z['ys_up'], rtol=1e-9)
    np.testing.assert_allclose(Fs2, z['Fs2'], rtol=0, atol=1e-9 * np.abs(z['Fs2']).max())


def test_elliptical_slice_restatement_matches_reference(golden_dir):
    """sliceSample.py:15-74 run by the reference (seeded) vs the oracle restatement given the same nu"""
    z = _load(golden_dir, 'elliptical_n40.npz')
    np.random.seed(int(z['seed']))
    f1 = sds_ref.elliptical_slice(z['f0'].copy(), z['x'], z['y'], z['hyp'])      # literal: draws nu itself
    tail = np.random.uniform(size=3)
    np.testing.assert_allclose(f1, z['f1'], rtol=0, atol=1e-10)
    np.testing.assert_array_equal(tail, z['rng_tail'])


def test_oracle_refuses_a_zero_likelihood_start():
    """guard added to the restatement (not in the reference): centred targets fall outside the truncation bounds, the
    reference would spin forever in its `while True`; the oracle raises so no test can hang on it"""
    rng = np.random.RandomState(0)
    x = np.arange(12.).reshape(-1, 1)
    y = rng.uniform(10, 90, 12)
    y = y - y.mean()
    with pytest.raises(ValueError, match='zero likelihood'):
        sds_ref.surrogate_slice_sampling(np.zeros(12), x, y, np.array([1., 10., 1.2]), np.array([10., 10., 5.]), literal=False)
    with pytest.raises(ValueError, match='zero likelihood'):
        sds_ref.elliptical_slice(np.zeros(12), x, y, np.array([1., 10., 1.2]), nu=np.ones(12))
