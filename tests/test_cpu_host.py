"""CPU-side checks: the C-ABI library loads and exports every symbol include/gpmc.h declares, host logic of the
mirrors, and the N > 1 chain-gather path over gloo with world_size 2.  No GPU compute here."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    txt = open(os.path.join(ROOT, 'include', 'gpmc.h')).read()
    txt = re.sub(r'/\*.*?\*/', '', txt, flags=re.S)
    return sorted(set(re.findall(r'\b(gpmc_[a-z0-9_]+)\s*\(', txt)))


def test_library_builds_loads_and_exports_header_symbols():
    import __graft_entry__ as ge
    ge.build()
    from gpmc import _lib
    lib = _lib.lib()
    names = _declared_symbols()
    assert len(names) >= 30
    for nm in names:
        assert hasattr(lib, nm), 'include/gpmc.h declares %s but libgpmc.so does not export it' % nm
    assert lib.gpmc_version() == 100
    assert lib.gpmc_pad(200) == 256 and lib.gpmc_pad(8192) == 8192
    assert lib.gpmc_potrf_scratch_bytes(256) > 0
    assert lib.gpmc_plan_workspace_bytes(8192, 2, 8) > 8192 * 8192 * 8
    assert lib.gpmc_plan_workspace_bytes(0, 2, 8) == 0
    # bad arguments are refused with a message, without touching a device
    assert lib.gpmc_set_option(b'no_such_option', 1) != 0 and b'unknown option' in lib.gpmc_last_error()
    assert lib.gpmc_set_option(b'outer_nb', 100) != 0
    assert lib.gpmc_potrf_f64(None, 128, 128, 128, None, None, None) < 0


def test_abi_under_host_sanitizers():
    """the host side of the library under AddressSanitizer + UBSan (``make SANITIZE=1``; GPU
    sanitizers are not available on this pool).  Every entry point that needs no device is driven in a child process
    with the ASan runtime preloaded: version / sizes, the option table (good, bad and unknown names, i.e. the
    vsnprintf error paths), argument validation of every factorisation / plan entry point, gpmc_last_error."""
    from gpmc import _lib
    if not os.path.exists(_lib.ASAN_LIB_PATH):
        _lib.build_sanitized()
    import glob
    rt = glob.glob('/opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so')
    if not rt:
        pytest.skip('ASan runtime not found')
    script = r"""
import ctypes, sys
L = ctypes.CDLL(sys.argv[1])
L.gpmc_last_error.restype = ctypes.c_char_p
L.gpmc_potrf_scratch_bytes.restype = ctypes.c_size_t
L.gpmc_plan_workspace_bytes.restype = ctypes.c_size_t
assert L.gpmc_version() == 100 and L.gpmc_pad(200) == 256 and L.gpmc_pad(8192) == 8192
assert L.gpmc_potrf_scratch_bytes(8192) > 2 << 20
assert L.gpmc_plan_workspace_bytes(8192, 2, 8) > 8192 * 8192 * 8 and L.gpmc_plan_workspace_bytes(0, 2, 8) == 0
for name in (b'outer_nb', b'lookahead', b'tail_m', b'dag_workers', b'tail_workers', b'la_dag', b'replay_form', b'use_graph',
             b'tail_acc16', b'tail_prefetch_t', b'dag_prefetch', b'tail_m_f32', b'kbuild_split', b'result_poll_us', b'tail_m_tri', b'tail_whole_m_tri'):
    v = L.gpmc_get_option(name); assert L.gpmc_set_option(name, v) == 0 and L.gpmc_get_option(name) == v, name
assert L.gpmc_get_option(b'tail_workers') == 0      # 0 = auto (worker count by the launch's bulk-to-chain ratio) survives a set
assert L.gpmc_set_option(b'outer_nb', 100) != 0 and b'multiple of 128' in L.gpmc_last_error()
assert L.gpmc_set_option(b'tail_m', 100) != 0 and L.gpmc_set_option(b'no_such_option_' + b'x' * 600, 1) != 0
assert L.gpmc_set_option(None, 1) != 0 and L.gpmc_get_option(None) == -1 and L.gpmc_get_option(b'nope') == -1
P = ctypes.c_void_p
assert L.gpmc_potrf_f64(None, 256, 256, 256, None, None, None) < 0 and b'bad argument' in L.gpmc_last_error()
assert L.gpmc_potrf_tri_f64(None, 256, 256, 256, 128, None, None, None) < 0
assert L.gpmc_kbuild_f64(0, None, 0, 0, 0, None, 0, 0, None, None, 0, 0, 0, 0, None) != 0
assert L.gpmc_syrk_tri_f64(None, 0, ctypes.c_double(0.), None, 0, 0, 0, None) < 0
assert L.gpmc_syrk_tri2_f64(None, 0, ctypes.c_double(0.), None, None, 0, 0, 0, None) < 0
assert L.gpmc_trsm_rows_f64(None, 0, 0, None, 0, 0, None) < 0 and L.gpmc_gemm_nt_f64(None, 0, None, 0, None, 0, 0, 0, 0, None) < 0
h = P()
assert L.gpmc_plan_create(ctypes.byref(h), 7, 10, 2, 8, None, 0, None) < 0      # bad kernel id
assert L.gpmc_plan_create(ctypes.byref(h), 0, 10, 99, 8, None, 0, None) < 0     # d too large
assert L.gpmc_plan_create(None, 0, 10, 2, 8, None, 0, None) < 0
assert L.gpmc_plan_destroy(None) == 0 and L.gpmc_profile_read(None) != 0
print('sanitized ABI ok')
"""
    env = dict(os.environ, LD_PRELOAD=rt[0], ASAN_OPTIONS='detect_leaks=0:abort_on_error=1', UBSAN_OPTIONS='halt_on_error=1')
    out = subprocess.run([sys.executable, '-c', script, _lib.ASAN_LIB_PATH], env=env, capture_output=True, timeout=240)
    assert out.returncode == 0 and b'sanitized ABI ok' in out.stdout, (out.stdout[-2000:], out.stderr[-4000:])


def test_product_path_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip('GPU present')
    from gpmc import _lib, core, covK, tools
    with pytest.raises(_lib.GpmcError):
        covK.RBF(0., 0.).getCovMatrix(x=np.zeros((4, 1)), mode='train')
    with pytest.raises(_lib.GpmcError):
        tools.jitchol(np.eye(3))
    with pytest.raises(_lib.GpmcError):
        core.LogMarginalPlan(np.zeros((4, 1)), np.zeros(4))


def test_host_helpers_match_oracle():
    from gpmc import covK, likK, sliceSample
    from oracle import sds_ref
    from oracle.kcgp_ref import likK as olik
    hyp = np.array([0.35, 2.0, 0.2])
    k, sn = covK.kernel_from_hyp(hyp)
    kp = k.kparams()
    assert kp[0] == np.exp(2. * np.log(2.0)) and kp[2] == np.exp(np.log(0.35)) and sn == 0.2
    assert covK.s_diag_scalar(kp[0], sn) == sds_ref.s_diag(np.array([kp[0]]), sn)[0]
    ka = covK.MaternARD52(np.log([1., 2., 3.]), np.log(1.5)).kparams(0.5)
    np.testing.assert_allclose(ka, [2.25, 0.5, 1., 2., 3.])
    rng = np.random.RandomState(0)
    y, mu = rng.uniform(-40, 40, 50), rng.uniform(-40, 40, 50)
    a = likK.TruncatedGauss2(upper=50., lower=-50., log_sigma=np.log(1.3))
    b = olik.TruncatedGauss2(upper=50., lower=-50., log_sigma=np.log(1.3))
    assert a.evaluate(y=y, mu=mu) == b.evaluate(y=y, mu=mu)
    for u, v in zip(a.evaluate(mu=mu, s2=np.full(50, 0.4)), b.evaluate(mu=mu, s2=np.full(50, 0.4))):
        np.testing.assert_array_equal(u, v)
    x = np.array([1., 10., 1.2])
    np.testing.assert_array_equal(sliceSample.log_gamma(x, sliceSample.PRIOR_K, sliceSample.PRIOR_THETA, True)[0],
                                  sds_ref.log_gamma(x, sds_ref.PRIOR_K, sds_ref.PRIOR_THETA, True)[0])


_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, 'gaussianprocess-mcmc_amd'))
from gpmc import chains

def fake_chain(x, y, iters, hyp0):
    # stands in for the GPU sampler: consumes the rank's seeded legacy stream
    h = np.random.uniform(size=(3, iters))
    f = np.random.standard_normal((len(y), iters))
    return f, h, np.ones(iters, dtype=np.int64)

x = np.zeros((5, 1)); y = np.zeros(5)
out = chains.run_chains(x, y, 4, (1., 10., 1.2), seed_base=124, gather_f=True, chain_fn=fake_chain)
assert out['world'] == 2 and out['histHyp'].shape == (2, 3, 4) and out['histF'].shape == (2, 5, 4)
for r in range(2):
    np.random.seed(124 + r)
    h = np.random.uniform(size=(3, 4)); f = np.random.standard_normal((5, 4))
    assert np.array_equal(out['histHyp'][r], h) and np.array_equal(out['histF'][r], f)
print('rank', out['rank'], 'ok')
'''


def test_truncated_gauss_against_scipy_truncnorm():
    """An independent pin for likK.TruncatedGauss2 (a14; the class is absent from the reference, semantics inferred
    from its call sites): log-density, mean and the 2.5 % / 97.5 % quantiles of a Gaussian(mu, sqrt(sn^2 + s2))
    truncated to [lower, upper] against scipy.stats.truncnorm, and -inf outside the bounds."""
    from scipy.stats import truncnorm
    from gpmc import likK
    rng = np.random.RandomState(1)
    lo, up, sn = -42.5, 57.5, 1.7
    mu = rng.uniform(-45, 60, 200)
    y = np.clip(mu + sn * rng.standard_normal(200), lo + 1e-3, up - 1e-3)
    lik = likK.TruncatedGauss2(upper=up, lower=lo, log_sigma=np.log(sn))
    a, b = (lo - mu) / sn, (up - mu) / sn
    ref = truncnorm.logpdf(y, a, b, loc=mu, scale=sn).sum()
    assert abs(lik.evaluate(y=y, mu=mu) - ref) <= 1e-10 * abs(ref)
    y_out = y.copy(); y_out[7] = up + 0.5
    assert lik.evaluate(y=y_out, mu=mu) == -np.inf
    s2 = rng.uniform(0.1, 4.0, 200)
    sd = np.sqrt(sn ** 2 + s2)
    a, b = (lo - mu) / sd, (up - mu) / sd
    ymu, lw, hi = lik.evaluate(mu=mu, s2=s2)
    np.testing.assert_allclose(ymu, truncnorm.mean(a, b, loc=mu, scale=sd), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(lw, truncnorm.ppf(0.025, a, b, loc=mu, scale=sd), rtol=1e-8, atol=1e-8)
    np.testing.assert_allclose(hi, truncnorm.ppf(0.975, a, b, loc=mu, scale=sd), rtol=1e-8, atol=1e-8)
    ref2 = truncnorm.logpdf(y, a, b, loc=mu, scale=sd).sum()
    assert abs(lik.evaluate(y=y, mu=mu, s2=s2) - ref2) <= 1e-10 * abs(ref2)


def test_chain_gather_world_size_2_gloo(tmp_path):
    """N > 1 path: one chain per rank, seeds base + rank, one all_gather of the histories (gloo, CPU)."""
    script = tmp_path / 'worker.py'
    script.write_text(_WORKER % {'root': ROOT})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29531', str(script)],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count('ok') == 2


def test_single_process_gather_is_identity():
    from gpmc import chains
    h = np.arange(12.).reshape(3, 4)
    a, b = chains.gather_histories(h, None)
    assert a.shape == (1, 3, 4) and b is None and np.array_equal(a[0], h)


# ----------------------------------------------------------------------- cross-validation harness
def _cv_data(n=53, seed=0):
    rng = np.random.RandomState(seed)
    return np.hstack([rng.uniform(0, 100, (n, 1)), np.arange(float(n)).reshape(-1, 1)])


def test_get_fold_data_matches_oracle_and_partitions_the_data():
    """framework.py:124-147: train / test split per (fold, gap, window), including folds whose last test ids run past
    the data and window sizes for which no test id exists at all."""
    from gpmc import framework
    from oracle import framework_ref as fr
    for n in (53, 60, 7):
        F = framework.Framework(_cv_data(n))
        for gap, win in ((2, 4), (3, 5), (1, 1), (5, 2), (7, 50), (4, 0)):
            for fold in range(gap + win):
                a = F.getFoldData(fold, gap, win)
                b = fr.get_fold_data(F.x, F.y, fold, gap, win)
                for u, v in zip(a, b):
                    np.testing.assert_array_equal(u, v)
                tr_x, tr_y, te_x, te_y, te_id = a
                assert tr_y.ndim == 1 and te_y.ndim == 1 and tr_x.shape == (len(tr_y), 1)
                assert len(tr_y) + len(np.unique(te_id)) == n                 # every point is train xor test
                assert np.array_equal(te_x[:, 0], te_id.astype(float))        # x = arange(n) here
                if n // (gap + win) > 0 and gap > 0:
                    assert te_id[0] == fold                                   # first masked block starts at `fold`
    F = framework.Framework(_cv_data(60))
    _, _, _, _, te = F.getFoldData(5, 2, 4)           # the fifth and last fold of gap 2 / window 4: ids 5,6, 11,12, ...
    assert list(te[:4]) == [5, 6, 11, 12] and te.max() == 59 and len(te) == 19          # id 60 falls off the end


def test_scale_options_and_selected_samples():
    from gpmc import framework
    from oracle import framework_ref as fr
    d = _cv_data(40, seed=3)
    for opt in (None, 'ctr', 'std'):
        F = framework.Framework(d)
        F.setScaleOpt(opt)
        F.getScaleData()
        np.testing.assert_array_equal(F.y, fr.scale_data(d[:, :1], opt))
    assert abs(F.y.mean()) < 1e-12 and abs(F.y.std() - 1.0) < 1e-12
    assert framework.selected_samples(1000) == list(range(899, 1000, 10)) == fr.selected_samples(1000)
    assert framework.selected_samples(30) == [26] and framework.selected_samples(20) == [17]
    with pytest.raises(Exception):
        framework.Framework(None)
    with pytest.raises(NotImplementedError):
        framework.crossValid(d, 4, [2]).execute(updOpt='opt')


_CV_WORKER = r'''
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(%(root)r, 'gaussianprocess-mcmc_amd'))
from gpmc import chains, framework

rank, world = chains.init_distributed()
rng = np.random.RandomState(0)
data = np.hstack([rng.uniform(0, 100, (53, 1)), np.arange(53.).reshape(-1, 1)])
ran = []

def fake_fold(gap, fold, iters):
    # stands in for chain + predictive on the GPU: a number that depends on the fold's seeded stream only
    ran.append((gap, fold))
    return float(np.random.uniform()) + 10 * gap + fold, np.zeros((53, iters)), np.zeros((3, iters))

cv = framework.crossValid(data, 4, [2, 3])
res = cv.execute(updOpt='mcmcSml', iterMCMC=5, fold_fn=fake_fold, outdir=%(out)r)
tasks = [(g, f) for g in (2, 3) for f in range(g + 4)]
assert ran == [t for i, t in enumerate(tasks) if i %% world == rank], ran      # round-robin shard, in order
for i, (g, f) in enumerate(tasks):
    np.random.seed(124 + i)
    want = float(np.random.uniform()) + 10 * g + f
    assert res[g][f] == want, (g, f, res[g][f], want)
assert sorted(res) == [2, 3] and len(res[2]) == 6 and len(res[3]) == 7
print('rank', rank, 'ok')
'''


def test_cross_valid_fold_fanout_world_size_2_gloo(tmp_path):
    """(gap, fold) tasks are dealt round robin to ranks, seeded per task, and ONE all_gather returns every fold's
    number to every rank; rank 0 writes llkGap<gap>.csv in the reference's layout (gloo, CPU, no GPU compute)."""
    script = tmp_path / 'cv_worker.py'
    out = tmp_path / 'out'
    script.write_text(_CV_WORKER % {'root': ROOT, 'out': str(out)})
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', CUDA_VISIBLE_DEVICES='', HIP_VISIBLE_DEVICES='')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29533', str(script)],
                       env=env, capture_output=True, text=True, timeout=240)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count('ok') == 2
    rows = (out / 'llkGap3.csv').read_text().strip().splitlines()
    assert rows[0] == 'gap,0,1,2,3,4,5,6' and rows[1].startswith('3,') and len(rows[1].split(',')) == 8
    assert (out / 'hypGap2.csv').read_text().splitlines()[0] == 'll,sf2,sn'


def test_bench_gpus_flag_spawns_its_own_ranks():
    """`python bench.py --gpus 2` with no launcher around it must come back as a 2-rank job (round 1 parsed the flag
    and ignored it).  CPU rehearsal: gloo + the stub plan (GPMC_BENCH_STUB=1: no GPU work, plumbing only)."""
    import json
    import subprocess
    env = dict(os.environ, GPMC_DIST_BACKEND='gloo', GPMC_BENCH_STUB='1')
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT'):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--steps', '3', '--warmup', '1',
                          '--n', '64'], env=env, capture_output=True, timeout=240)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1                                   # ONE JSON line, from rank 0
    doc = json.loads(lines[0])
    assert doc['n_gpus'] == 2 and doc['steps'] == 3 and doc['scaling'] == 'weak'
    assert doc['value'] > 0 and abs(doc['value'] - 2 * 3 / (doc['ms_per_step'] * 3e-3)) < 1e-6 * doc['value']
    assert 'STUB' in doc['data'] and doc['cpu_baseline'] is None


def test_bench_under_a_launcher_keeps_working():
    """the torch.distributed.run form the driver uses for N > 1 must not spawn again"""
    import json
    import subprocess
    env = dict(os.environ, GPMC_DIST_BACKEND='gloo', GPMC_BENCH_STUB='1')
    out = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                          '--master-addr', '127.0.0.1', '--master-port', '29611', os.path.join(ROOT, 'bench.py'),
                          '--gpus', '2', '--steps', '2', '--warmup', '1'], env=env, capture_output=True, timeout=240)
    assert out.returncode == 0, out.stderr.decode()[-2000:]
    lines = [ln for ln in out.stdout.decode().splitlines() if ln.startswith('{')]
    assert len(lines) == 1 and json.loads(lines[0])['n_gpus'] == 2


def test_timeout_codes_and_shared_card_detection(monkeypatch):
    """host logic of round 3: the tile-DAG kernel's per-cause timeout codes (csrc/tail.h tail_timeout_code) decode to
    (cause, tile column) and raise GpmcError naming the wait; ordinary LAPACK info passes through; shared-card detection
    reads LOCAL_WORLD_SIZE only and never guesses from WORLD_SIZE."""
    from gpmc import _lib, chains
    assert _lib.decode_timeout(0) is None and _lib.decode_timeout(17) is None and _lib.decode_timeout(-7) is None
    assert _lib.decode_timeout(-1001) == (0, -1)
    assert _lib.decode_timeout(-1010042) == (1, 42) and _lib.decode_timeout(-1030000) == (3, 0)
    assert _lib.check_info(0) == 0 and _lib.check_info(4097.0) == 4097
    for code, word in ((-1010042, 'GEMM'), (-1020007, 'operand tile'), (-1030127, 'diagonal factor'), (-1001, 'another workgroup')):
        with pytest.raises(_lib.GpmcError, match='timed out') as ei:
            _lib.check_info(code)
        assert word in str(ei.value) and 'shared_card' in str(ei.value)
    monkeypatch.delenv('LOCAL_WORLD_SIZE', raising=False)
    monkeypatch.setenv('WORLD_SIZE', '16')
    assert chains.ranks_share_a_card() is False          # multi-node launch without LOCAL_WORLD_SIZE: nothing is guessed
