"""SURVEY.md 8(f) "next" rows on the device: inf_mcmc, elliptical_slice, CSV outputs.  -m gpu only."""
import csv
import os

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

from oracle import sds_ref                      # noqa: E402  (checker only)


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name))


def test_inf_mcmc_matches_reference_fixture(golden_dir):
    """the reference's own inf_mcmc output (tests/golden/inf_mcmc_n60.npz) vs the device path"""
    from gpmc import covK, gpK, likK, sliceSample
    z = _load(golden_dir, 'inf_mcmc_n60.npz')
    hyp = z['hyp']
    model = gpK.GPR()
    model.setData(z['x'], z['y'])
    model.xs = z['xs']
    model.setPrior(kernel=covK.RBF(np.log(hyp[0]), np.log(hyp[1])))
    model.likfunc = likK.TruncatedGauss2(upper=100. - z['y'].mean(), lower=0. - z['y'].mean(), log_sigma=np.log(hyp[2]))
    ym, lw, up, Fs2 = sliceSample.inf_mcmc(z['f'], model)
    np.testing.assert_allclose(ym, z['ym'], rtol=1e-8)
    np.testing.assert_allclose(lw, z['ys_lw'], rtol=1e-7)
    np.testing.assert_allclose(up, z['ys_up'], rtol=1e-7)
    np.testing.assert_allclose(Fs2, z['Fs2'], rtol=0, atol=1e-8 * np.abs(z['Fs2']).max())


def test_inf_mcmc_larger_vs_oracle():
    from gpmc import covK, gpK, likK, sliceSample
    from oracle.kcgp_ref import covK as ocov, likK as olik
    rng = np.random.RandomState(3)
    n, ns, nsamp = 700, 150, 9
    x = rng.uniform(0, 10, (n, 2)); xs = rng.uniform(0, 10, (ns, 2)); y = rng.uniform(20, 80, n)
    f = rng.standard_normal((n, nsamp)) * 3
    model = gpK.GPR(); model.setData(x, y); model.xs = xs
    model.setPrior(mean=gpK.Const(0.5), kernel=covK.RBF(np.log(1.4), np.log(6.0)))
    model.likfunc = likK.TruncatedGauss2(upper=100. - y.mean(), lower=-y.mean(), log_sigma=np.log(1.1))
    got = sliceSample.inf_mcmc(f, model)
    om = sds_ref.Model(x, y, xs, ocov.RBF(np.log(1.4), np.log(6.0)),
                       olik.TruncatedGauss2(upper=100. - y.mean(), lower=-y.mean(), log_sigma=np.log(1.1)),
                       sds_ref.ConstMean(0.5))
    ref = sds_ref.inf_mcmc(f, om)
    for a, b in zip(got, ref):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-7 * max(1.0, np.abs(b).max()))


def test_elliptical_slice_given_nu_matches_reference_fixture(golden_dir):
    """angle-bracket loop draw for draw: with the fixture's nu the device-side function returns the reference's f"""
    from gpmc import sliceSample
    z = _load(golden_dir, 'elliptical_n40.npz')
    np.random.seed(int(z['seed']))
    np.random.multivariate_normal(np.zeros(40), np.eye(40), 1)          # advance the stream as the reference's draw did
    f1 = sliceSample.elliptical_slice(z['f0'].copy(), z['x'], z['y'], z['hyp'], nu=z['nu'])
    tail = np.random.uniform(size=3)
    np.testing.assert_allclose(f1, z['f1'], rtol=0, atol=1e-10)
    np.testing.assert_array_equal(tail, z['rng_tail'])


def test_elliptical_slice_device_draw_has_prior_covariance():
    """nu = chol(K) z on the device: the update keeps f on the prior ellipse and moves it"""
    from gpmc import sliceSample
    rng = np.random.RandomState(5)
    n = 120
    x = np.sort(rng.uniform(0, 10, (n, 1)), axis=0)
    y = np.clip(50 + 10 * np.sin(x[:, 0]) + rng.standard_normal(n), 1, 99)
    f = y - y.mean()
    np.random.seed(9)
    f1 = sliceSample.elliptical_slice(f, x, y, np.array([1.0, 10.0, 1.2]))
    assert f1.shape == (n,) and np.all(np.isfinite(f1)) and not np.allclose(f1, f)


def test_csv_outputs_have_reference_layout(tmp_path):
    from gpmc import framework
    histHyp = np.arange(6.).reshape(3, 2)
    histF = np.arange(8.).reshape(4, 2)
    x, y = np.arange(4.), np.arange(4.) * 2
    framework.output(x, y, gap=0, histHyp=histHyp, histF=histF, llk=[0.1, 0.2], outdir=str(tmp_path))
    rows = list(csv.reader(open(tmp_path / 'hypGap0.csv')))
    assert rows[0] == ['ll', 'sf2', 'sn'] and len(rows) == 4
    rows = list(csv.reader(open(tmp_path / 'fGap0.csv')))
    assert rows[0] == ['1', '2', 'x', 'y'] and len(rows) == 5 and rows[1][2:] == ['0.0', '0.0']
    framework.outputDemoRes(x, y, histF, histHyp, outdir=str(tmp_path))
    rows = list(csv.reader(open(tmp_path / 'demo_hyp.csv')))
    assert rows[0] == ['ll', 'sf2', 'sn'] and len(rows) == 3
