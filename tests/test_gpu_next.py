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
    """nu = chol(K) z on the device is equal IN DISTRIBUTION to the reference's multivariate_normal(0, K) (an SVD of K,
    sliceSample.py:41): over 4000 draws at n = 16 the sample mean is 0 and the sample covariance is K, entry by entry,
    within 5 standard errors (se of an entry of the sample covariance of a Gaussian: sqrt((K_ii K_jj + K_ij^2) / M));
    and one update stays on the prior ellipse and moves."""
    from gpmc import sliceSample
    from oracle.kcgp_ref import covK as ocov
    rng = np.random.RandomState(5)
    n, M = 16, 4000
    x = np.sort(rng.uniform(0, 10, (n, 1)), axis=0)
    hyp = np.array([1.0, 10.0, 1.2])
    K = ocov.RBF(np.log(hyp[0]), np.log(hyp[1])).getCovMatrix(x=x, mode='train')
    np.random.seed(11)
    nu = sliceSample.draw_prior(x, hyp, size=M)
    assert nu.shape == (M, n)
    se_mean = np.sqrt(np.diag(K) / M)
    assert np.all(np.abs(nu.mean(axis=0)) <= 5 * se_mean)
    S = nu.T @ nu / M
    se = np.sqrt((np.outer(np.diag(K), np.diag(K)) + K ** 2) / M)
    assert np.all(np.abs(S - K) <= 5 * se), float((np.abs(S - K) / se).max())
    y = np.clip(50 + 10 * np.sin(x[:, 0]) + rng.standard_normal(n), 1, 99)
    f = y - y.mean()
    np.random.seed(9)
    f1 = sliceSample.elliptical_slice(f, x, y, hyp)
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


# ----------------------------------------------------------------------- cross-validation harness (8(f) rank 4)
def _cv_series(n=48, seed=11):
    """a smooth bounded 1-D series in the reference's data layout [target | x]"""
    rng = np.random.RandomState(seed)
    x = np.arange(float(n))
    y = 50. + 25. * np.sin(x / 6.) + 3. * rng.standard_normal(n)
    return np.hstack([np.clip(y, 1., 99.).reshape(-1, 1), x.reshape(-1, 1)])


@pytest.mark.parametrize('pairing', ['literal', 'matched'])
def test_cross_valid_matches_oracle(tmp_path, pairing):
    """crossValid.execute (framework.py:194-248) on the device vs the CPU restatement, same global seed: six folds of
    gap 2 / window 4, each a 20-iteration chain from the framework's initial point plus the predictive at the
    held-out points.  The chains must take the same accept / shrink decisions for the streams to stay aligned; the
    latent f agrees to ~1e-2 (R is ill-conditioned, see test_gpu_sds.py), hence the tolerance on the fold numbers."""
    from gpmc import framework
    from oracle import framework_ref as fr
    data = _cv_series()
    np.random.seed(7)
    cv = framework.crossValid(data, 4, [2])
    got = cv.execute(updOpt='mcmcSml', iterMCMC=20, pairing=pairing, outdir=str(tmp_path))
    want = fr.cross_valid(data, 4, [2], 20, pairing=pairing, seed=7)
    assert sorted(got) == [2] and len(got[2]) == 6 and cv.testLLK == got
    np.testing.assert_allclose(got[2], want[2], rtol=2e-2)
    assert np.all(np.isfinite(got[2]))
    # CSVs of the gap in the reference's layout: the last fold's histories against ITS training x, y
    rows = list(csv.reader(open(os.path.join(str(tmp_path), 'llkGap2.csv'))))
    assert rows[0] == ['gap', '0', '1', '2', '3', '4', '5'] and float(rows[1][0]) == 2
    np.testing.assert_allclose([float(v) for v in rows[1][1:]], got[2])
    frow = list(csv.reader(open(os.path.join(str(tmp_path), 'fGap2.csv'))))
    assert frow[0] == [str(i) for i in range(1, 21)] + ['x', 'y'] and len(frow) - 1 == cv.x.shape[0] == 48 - 15   # fold 5 masks 5,6,11,12,...,47 (48 is past the end)
    hrow = list(csv.reader(open(os.path.join(str(tmp_path), 'hypGap2.csv'))))
    assert hrow[0] == ['ll', 'sf2', 'sn'] and len(hrow) == 21


def test_cross_valid_seed_per_fold_is_partition_independent():
    """with seed_base every fold draws from its own stream: running a fold alone gives the number execute reports"""
    from gpmc import framework
    data = _cv_series(40, seed=5)
    cv = framework.crossValid(data, 3, [2])
    res = cv.execute(updOpt='mcmcSml', iterMCMC=12, seed_base=300, outdir=None)
    np.random.seed(300 + 3)                                   # task 3 = (gap 2, fold 3)
    llk, _, _ = cv.runFold(2, 3, 12)
    assert llk == res[2][3]


def test_single_run_writes_reference_csvs(tmp_path):
    from gpmc import framework
    data = _cv_series(32, seed=2)
    np.random.seed(1)
    sr = framework.singleRun(data)
    histF, histHyp = sr.execute(updOpt='mcmcSml', iterMCMC=6, outdir=str(tmp_path))
    assert histF.shape == (32, 6) and histHyp.shape == (3, 6)
    h = list(csv.reader(open(os.path.join(str(tmp_path), 'hypGap0.csv'))))
    assert h[0] == ['ll', 'sf2', 'sn'] and len(h) == 7
    np.testing.assert_allclose([float(v) for v in h[1]], histHyp[:, 0])
    with pytest.raises(NotImplementedError):
        sr.execute(updOpt='opt')


def test_zero_likelihood_start_raises_instead_of_spinning(tmp_path):
    """Centred targets leave the truncation bounds [0, 100] - mean(y): the current state has likelihood zero and the
    reference's `while True` (sliceSample.py:129-163) can never return.  The device sampler must fail loudly."""
    from gpmc import framework, sliceSample
    data = _cv_series(32, seed=2)
    sr = framework.singleRun(data)
    sr.setScaleOpt('ctr')
    with pytest.raises(ValueError, match='zero likelihood'):
        sr.execute(updOpt='mcmcSml', iterMCMC=3, outdir=str(tmp_path))
    assert abs(sr.y.mean()) < 1e-12
    y = sr.y.reshape(-1)
    with pytest.raises(ValueError, match='zero likelihood'):
        sliceSample.elliptical_slice(np.zeros_like(y), sr.x, y, np.array([1., 10., 1.2]))


# ----------------------------------------------------------------------- chain-parallel driver with real device chains
_RANKS_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, 'gaussianprocess-mcmc_amd'))
sys.path.insert(0, %(root)r)
from gpmc import chains, framework
from oracle import sds_ref

x, y = sds_ref.synthetic_demo_data(n=160, seed=124)
y = np.clip(y, 1., 99.)
out = chains.run_chains(x, y, 8, framework.DEMO_HYP0, seed_base=500, gather_f=True)
assert out['world'] == 2 and out['histHyp'].shape == (2, 3, 8) and out['histF'].shape == (2, 160, 8)
# every rank checks BOTH gathered chains against chains it runs alone with the same seeds
for r in range(2):
    np.random.seed(500 + r)
    histF, histHyp, _ = framework.run_chain(x, y, 8, framework.DEMO_HYP0)
    assert np.array_equal(out['histHyp'][r], histHyp), r
    assert np.array_equal(out['histF'][r], histF), r
assert not np.array_equal(out['histHyp'][0], out['histHyp'][1])
print('rank', out['rank'], 'ok', flush=True)
'''


def test_two_ranks_run_independent_device_chains(tmp_path):
    """SURVEY.md 8(e) end to end on the device: two ranks (sharing this box's one GPU, process group over gloo because
    RCCL wants one device per rank) each run their own seeded chain, one all_gather returns both histories to both
    ranks, and each history is bit-identical to the same chain run alone."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'ranks_worker.py'
    script.write_text(_RANKS_WORKER % {'root': root})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', GPMC_DIST_BACKEND='gloo')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29537', str(script)],
                       env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.count('ok') == 2


@pytest.mark.parametrize('n', [8192, 12288])
def test_two_processes_share_one_card_at_full_size(n):
    """Two ranks of the bench on ONE card (gloo; the driver's real launches have a card per rank), once at N = 8192 and
    once at N = 12288 -- the sizes where round 2 recorded info = -1001 with 512-column panels (14 of 24 runs at 8192,
    1 of 3 at 12288 even with 384-column panels).  Cause: a look-ahead panel launch gated on the flags of a GEMM that has
    not been dispatched holds CU slots while it spins, and two processes' gated launches (2 x 264 workgroups) hold all
    512, so neither GEMM can place a block.  ``chains.init_distributed`` now sets ``shared_card`` when ranks outnumber
    cards and the hand-over becomes stream-ordered (csrc/potrf.hip): no workgroup waits for an unscheduled kernel.  One
    run per size, on the default 512-column schedule; a timeout would name its wait (tests below)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GPMC_DIST_BACKEND='gloo')
    env.pop('RANK', None), env.pop('WORLD_SIZE', None), env.pop('LOCAL_RANK', None)
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '2', '--steps', '6', '--warmup', '2',
                        '--metric-only', '--n', str(n)], env=env, capture_output=True, text=True, timeout=290)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('{')][-1]
    d = json.loads(line)
    assert d['n_gpus'] == 2 and d['value'] > 0, d
    assert d['config']['shared_card'] == 1, d['config']


def test_plans_of_one_process_evaluate_concurrently_at_full_size():
    """k independent chains on one card, one process: three plans at N = 8192, each on its own stream, each strictly
    sequential, interleaved by the host.  With in-launch hand-overs this ran into the bounded wait (info = -1010025 with
    two such plans; -1020026 with one of them beside another plan's tail launch): waiting workgroups of two factorisations
    hold every CU slot and the GEMM they wait for cannot place a block.  The in-launch form is therefore used only while
    one workspace has live plans (gpmc_plan_handover_in_launch); the others' appearance switches every plan to the
    stream-ordered hand-over and makes it record again.  The values are those each plan gives alone, bit for bit."""
    import gc
    from gpmc import core
    gc.collect()
    n = 8192
    hyps = [[1.0, 10.0, 1.2], [2.0, 8.0, 1.0], [1.5, 9.0, 1.1]]
    rng = np.random.RandomState(100)
    first = core.LogMarginalPlan(rng.uniform(0, 10, (n, 2)), rng.standard_normal(n) * 5)
    alone = [[first(h) for h in hyps]]
    plans = [first]
    for c in (1, 2):
        rng = np.random.RandomState(100 + c)
        plans.append(core.LogMarginalPlan(rng.uniform(0, 10, (n, 2)), rng.standard_normal(n) * 5))
        alone.append([plans[-1](h) for h in hyps])
    assert not any(p.handover_in_launch for p in plans)
    for p in plans:
        p.eval_async(hyps[0], 0)
    for s in range(9):
        for c, p in enumerate(plans):
            got = p.result(0)[0]
            assert got == alone[c][s % 3], (c, s, got, alone[c][s % 3])
            p.eval_async(hyps[(s + 1) % 3], 0)
    for p in plans:
        p.result(0)
    plans[1].close(), plans[2].close()
    if first.handover_in_launch:       # (not if a plan of an earlier test is still alive)
        assert first(hyps[1]) == alone[0][1]


_NCCL_WORKER = r'''
import os, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, 'gaussianprocess-mcmc_amd'))
import numpy as np, torch, torch.distributed as dist
from gpmc import chains
rank, world = chains.init_distributed()
assert (rank, world) == (0, 1) and dist.is_initialized() and dist.get_backend() == 'nccl', dist.get_backend()
dev = torch.device('cuda', torch.cuda.current_device())
dist.barrier(device_ids=[dev.index])
t = torch.tensor([3.25], dtype=torch.float64, device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)
assert float(t.item()) == 3.25
rng = np.random.RandomState(5)
hh, hf = rng.standard_normal((3, 12)), rng.standard_normal((40, 12))
allh, allf = chains.gather_histories(hh, hf)            # CUDA-tensor all_gather over RCCL
assert allh.shape == (1, 3, 12) and np.array_equal(allh[0], hh) and np.array_equal(allf[0], hf)
v = np.array([1.0, np.nan, 2.5])
got = chains.gather_vector(v)
assert np.array_equal(got[[0, 2]], v[[0, 2]]) and np.isnan(got[1])
dist.destroy_process_group()
print('nccl ok', flush=True)
'''


def test_rccl_path_executes_on_one_gpu(tmp_path):
    """the RCCL branch of the chain driver (``backend='nccl', device_id=dev``, ``dist.barrier(device_ids=)``,
    CUDA-tensor ``all_reduce`` / ``all_gather``) run for real on this box's one GPU: GPMC_FORCE_DIST=1 makes
    ``init_distributed`` create the group for a single rank.  Then ``bench.py --gpus 1`` under the same flag -- its
    barrier / max-over-ranks all_reduce go through RCCL -- must print its usual line."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = tmp_path / 'nccl_worker.py'
    script.write_text(_NCCL_WORKER % {'root': root})
    env = dict(os.environ, GPMC_FORCE_DIST='1', WORLD_SIZE='1', RANK='0', LOCAL_RANK='0', LOCAL_WORLD_SIZE='1',
               MASTER_ADDR='127.0.0.1', MASTER_PORT='29541', HSA_ENABLE_IPC_MODE_LEGACY='0')
    env.pop('GPMC_DIST_BACKEND', None)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0 and 'nccl ok' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
    env['MASTER_PORT'] = '29543'
    r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '4', '--warmup', '1',
                        '--metric-only', '--n', '2048'], env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-4000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert d['n_gpus'] == 1 and d['value'] > 0 and d['config']['collective_backend'] == 'nccl', d['config']
