"""CPU restatement of the surrogate-data slice-sampling hot path.

TEST INFRASTRUCTURE ONLY (see ``oracle/__init__.py``).  Follows
``/root/reference/kcMCMC/sliceSample.py:76-232`` (sampler, auxiliary model,
priors), ``framework.py:59-77`` and ``demoRegression.py:15-32`` (chain drivers),
``demoRegression.py:116-131`` (synthetic generator).  Pinned against fixtures
produced by the reference's own ``sliceSample.py`` (``oracle/make_golden.py``).

Two forms of every dense step are kept:

* ``*_literal``  -- the reference's op sequence (dense diagonal S, ``np.linalg.solve``
  on a triangular matrix, ``np.linalg.inv`` twice): ~11.3 N^3 flops.  This is
  the "B-ref" CPU baseline and the form checked against the golden vectors.
* ``*_lean``     -- the algebraically identical triangular form (dpotrf, dtrsm,
  dsyrk-shaped product, dtrsv): 8 N^3/3 flops.  This is what the HIP path
  computes; it is checked against the literal form to a stated tolerance.
"""
import numpy as np
import scipy.linalg
import scipy.special

from .kcgp_ref import covK, likK, tools

LOG_2PI = np.log(2. * np.pi)

# hard-coded sampler constants of the reference, exposed as defaults
PRIOR_K = np.asarray([1., 3., 3.])          # sliceSample.py:124
PRIOR_THETA = np.asarray([1., 1.5, 3.])     # sliceSample.py:125
NOISE_FREEZE_ITERS = 500                    # sliceSample.py:128,133,151
R_JITTER = 1e-11                            # sliceSample.py:205
Y_UPPER, Y_LOWER = 100., 0.                 # sliceSample.py:114-115


# --------------------------------------------------------------------------- priors
def log_gamma(x, k, theta, invG):
    """Gamma log-pdf per coordinate, coordinate 2 inverse-Gamma when ``invG``.

    sliceSample.py:209-232.  Returns (logpdf, gradient).
    """
    x = np.asarray(x, dtype=np.float64)
    lg = (k - 1.) * np.log(x) - x / theta - k * np.log(theta) - np.log(scipy.special.gamma(k))
    gr = (k - 1.) / x - 1. / theta
    if invG:
        lg[2] = (np.log(theta[2] ** k[2]) - np.log(scipy.special.gamma(k[2]))
                 + (-k[2] - 1.) * np.log(x[2]) + (-theta[2] / x[2]))
        gr[2] = (-k[2] - 1.) / x[2] + theta[2] / (x[2] ** 2)
    return lg, gr


# --------------------------------------------------------------------------- S diagonal
def s_diag(Kii, sn):
    """Sii = 1/(((sn^2)^-1 + 1/Kii) - 1/Kii), clamped at 0  (sliceSample.py:184-190)."""
    kinv = 1. / Kii
    v1 = (sn ** 2) ** (-1) + kinv
    return np.maximum(1. / (v1 - kinv), 0.)


def draw_g(f, Sii):
    """g ~ N(f, diag S) on the global legacy ``np.random`` stream.

    The reference draws with ``multivariate_normal(f, S, 1)`` (sliceSample.py:194),
    an SVD of the dense diagonal S; for S = s*I that equals
    ``f + sqrt(s)*standard_normal(N)`` bit for bit and leaves the stream at the
    same position.  The golden fixtures record g explicitly.
    """
    return f + np.sqrt(Sii) * np.random.standard_normal(f.shape[0])


# --------------------------------------------------------------------------- auxiliary model
def aux_var_model_literal(f, K, sn, g=None):
    """Literal op sequence of sliceSample.py:165-207 (dense S, solve, inv)."""
    n = K.shape[0]
    Sii = s_diag(np.diagonal(K), sn)
    S = np.zeros_like(K)
    np.fill_diagonal(S, Sii)
    if g is None:
        g = np.random.multivariate_normal(f, S, 1).T.reshape((n,))
    L = tools.jitchol(K + S)
    V = np.linalg.solve(L, K)
    R = K - np.dot(V.T, V)
    m = np.dot(np.dot(R, np.linalg.inv(S)), g)
    C = tools.jitchol(R + np.eye(n) * R_JITTER)
    return g, K + S, m, C, L


def aux_var_model_lean(f, K, sn, g=None):
    """Same quantities through triangular kernels; returns (g, Sii, m, C, L)."""
    n = K.shape[0]
    Sii = s_diag(np.diagonal(K), sn)
    if g is None:
        g = draw_g(f, Sii)
    A = K.copy()
    A[np.diag_indices(n)] += Sii
    L = tools.jitchol(A)
    V = scipy.linalg.solve_triangular(L, K, lower=True)      # L^-1 K
    R = K - V.T @ V
    m = R @ (g / Sii)
    R[np.diag_indices(n)] += R_JITTER
    C = tools.jitchol(R)
    return g, Sii, m, C, L


# --------------------------------------------------------------------------- log-marginal
def log_marginal_literal(g, K_S, L):
    """curG / propG exactly as written at sliceSample.py:122,147 (``inv`` form)."""
    return -(np.dot(np.dot(g.T, np.linalg.inv(K_S)), g) / 2.
             + np.log(np.diag(L.T)).sum() + g.shape[0] * LOG_2PI / 2.)


def log_marginal_lean(g, L):
    """log N(g; 0, L L^T) with alpha = L^-1 g (one TRSV)."""
    a = scipy.linalg.solve_triangular(L, g, lower=True)
    return -(0.5 * np.dot(a, a) + np.log(np.diag(L)).sum() + g.shape[0] * LOG_2PI / 2.)


def gp_log_marginal(x, g, hyp, kernel='rbf'):
    """The metric's unit: K build + diag add + Cholesky + log N(g; 0, K + sn^2 I).

    hyp = (ell.., sf, sn) on the LINEAR scale (the sampler's convention).
    """
    hyp = np.asarray(hyp, dtype=np.float64)
    if kernel == 'rbf':
        K = covK.RBF(np.log(hyp[0]), np.log(hyp[1])).getCovMatrix(x=x, mode='train')
    elif kernel == 'matern52ard':
        K = covK.MaternARD52(np.log(hyp[:-2]), np.log(hyp[-2])).getCovMatrix(x=x, mode='train')
    else:
        raise ValueError(kernel)
    n = K.shape[0]
    K[np.diag_indices(n)] += s_diag(np.diagonal(K).copy(), hyp[-1])
    L = scipy.linalg.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    return log_marginal_lean(np.asarray(g, dtype=np.float64), L)


# --------------------------------------------------------------------------- sampler
def _check_finite_start(cur_llk):
    """NOT in the reference: with a non-finite current log-likelihood (a target outside the truncation bounds, e.g.
    centred targets) its `while True` loops (sliceSample.py:59-74,129-163) never accept and never return.  The
    restatement raises instead so that a test cannot hang; the product raises the same error."""
    if not np.isfinite(cur_llk):
        raise ValueError('current state has zero likelihood under the truncated noise model; the sampler cannot move')


def surrogate_slice_sampling(f, x, y, hyp, scale, iter=0, literal=True, stats=None):
    """One SDS update of (f, hyp) -- sliceSample.py:76-163, draw for draw.

    ``stats`` (optional dict) receives the number of proposals evaluated and the margin
    ``proposal - threshold`` of every try (how close each accept/shrink decision was).
    RNG order on the global legacy stream: N normals (g), 3 uniforms (bracket),
    1 uniform (slice height), then 3 uniforms per proposal.
    """
    aux = aux_var_model_literal if literal else aux_var_model_lean
    my = np.mean(y)
    K = covK.RBF(np.log(hyp[0]), np.log(hyp[1])).getCovMatrix(x=x, mode='train')
    g, K_S, m, C, L = aux(f, K, hyp[2])
    if literal:
        ita = np.linalg.solve(C, f - m)
    else:
        ita = scipy.linalg.solve_triangular(C, f - m, lower=True)

    v = np.random.uniform(low=0., high=scale)
    hyp_min = np.maximum(hyp - v, 0)
    hyp_max = hyp_min + scale

    lik = likK.TruncatedGauss2(upper=Y_UPPER - my, lower=Y_LOWER - my, log_sigma=np.log(hyp[2]))
    cur_llk = lik.evaluate(y=y - my, mu=f)
    _check_finite_start(cur_llk)
    curG = log_marginal_literal(g, K_S, L) if literal else log_marginal_lean(g, L)

    prior, _ = log_gamma(hyp, PRIOR_K, PRIOR_THETA, True)
    threshold = np.log(np.random.uniform()) + cur_llk + prior[1] + prior[0] + curG
    if iter >= NOISE_FREEZE_ITERS:
        threshold += prior[2]

    ntry = 0
    margins = []
    while True:
        ntry += 1
        prop_hyp = np.random.uniform(low=hyp_min, high=hyp_max)
        if iter < NOISE_FREEZE_ITERS:
            prop_hyp[2] = hyp[2]
        nK = covK.RBF(np.log(prop_hyp[0]), np.log(prop_hyp[1])).getCovMatrix(x=x, mode='train')
        g, K_S, m, C, L = aux(f, nK, prop_hyp[2], g=g)
        prop_f = np.dot(C, ita) + m
        lik.sn = prop_hyp[2]
        prop_llk = lik.evaluate(y=y - my, mu=prop_f)
        propG = log_marginal_literal(g, K_S, L) if literal else log_marginal_lean(g, L)
        pp, _ = log_gamma(prop_hyp, PRIOR_K, PRIOR_THETA, True)
        proposal = prop_llk + pp[1] + pp[0] + propG
        if iter >= NOISE_FREEZE_ITERS:
            proposal += pp[2]
        margins.append(proposal - threshold)
        if proposal > threshold and np.isfinite(proposal):
            if stats is not None:
                stats['proposals'] = ntry
                stats['margins'] = margins
            return prop_f, prop_hyp
        for i in range(3):
            if prop_hyp[i] < hyp[i]:
                hyp_min[i] = prop_hyp[i]
            else:
                hyp_max[i] = prop_hyp[i]


# --------------------------------------------------------------------------- chain drivers
def run_chain(x, y, iters, hyp0, scale=(10., 10., 5.), literal=True, iter0=0, margins=None):
    """Shape of Framework.runSimulMCMC (framework.py:59-77) / mcmcUpdate (demoRegression.py:15-32).

    Returns histF (n, iters), histHyp (3, iters), proposals (iters,); ``margins`` (optional list) receives, per
    iteration, the list of ``proposal - threshold`` values of its tries.
    """
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    hyp = np.asarray(hyp0, dtype=np.float64)
    f = np.zeros_like(y)
    histF = np.zeros((y.shape[0], iters))
    histHyp = np.zeros((3, iters))
    ntries = np.zeros(iters, dtype=np.int64)
    st = {}
    for i in range(iters):
        f, hyp = surrogate_slice_sampling(f, x, y, hyp, scale=np.asarray(scale, dtype=np.float64),
                                          iter=iter0 + i, literal=literal, stats=st)
        histF[:, i] = f
        histHyp[:, i] = hyp
        ntries[i] = st['proposals']
        if margins is not None:
            margins.append(list(st['margins']))
    return histF, histHyp, ntries


FRAMEWORK_HYP0 = (1., 10., 1.2)     # framework.py:63
DEMO_HYP0 = (0.35, 2.0, 0.2)        # demoRegression.py:15


def synthetic_demo_data(n=455, seed=124, ll=5.0, sf=20.0, sn=2.5, mean=91.1538461538):
    """demoRegression.py:116-131: x = arange(n), y = chol(K + sn^2 I) z + mean."""
    np.random.seed(seed)
    x = np.arange(0, n)
    K = covK.RBF(np.log(ll), np.log(sf)).getCovMatrix(x.reshape((n, 1)), mode='train')
    L = tools.jitchol(K + sn ** 2 * np.eye(n))
    z = np.random.normal(size=(n,))
    y = np.dot(L, z) + mean
    return x.astype(np.float64).reshape(n, 1), y


def bench_inputs(n, d, seed=124, truth=(1.0, 10.0, 1.2), npoints=16, scale=(10., 10., 5.)):
    """Synthetic workload of SURVEY.md 8(d): X ~ U(0,10)^(n x d), g = chol(K+sn^2 I) z,
    and ``npoints`` hyper-parameter evaluation points drawn from the sampler's bracket,
    floored at 0.05."""
    rng = np.random.RandomState(seed)
    X = rng.uniform(0., 10., (n, d))
    K = covK.RBF(np.log(truth[0]), np.log(truth[1])).getCovMatrix(x=X, mode='train')
    K[np.diag_indices(n)] += truth[2] ** 2
    L = scipy.linalg.cholesky(K, lower=True, overwrite_a=True, check_finite=False)
    g = L @ rng.standard_normal(n)
    del K, L
    h = np.asarray(truth)
    s = np.asarray(scale)
    pts = []
    for _ in range(npoints):
        lo = np.maximum(h - rng.uniform(0., s), 0.)
        pts.append(np.maximum(rng.uniform(lo, lo + s), 0.05))
    return X, g, np.asarray(pts)


# --------------------------------------------------------------------------- "next" rows
class ZeroMean(object):
    def getMean(self, x):
        return np.zeros((np.asarray(x).shape[0], 1))


class ConstMean(object):
    def __init__(self, c):
        self.c = float(c)

    def getMean(self, x):
        return np.full((np.asarray(x).shape[0], 1), self.c)


class Model(object):
    """the attributes inf_mcmc reads from a kcGP.gpK.GPR instance (sliceSample.py:246-256,262-265,279)"""

    def __init__(self, x, y, xs, covfunc, likfunc, meanfunc=None):
        self.x, self.y, self.xs = x, y, xs
        self.covfunc, self.likfunc = covfunc, likfunc
        self.meanfunc = meanfunc or ZeroMean()


def inf_mcmc(f, model, ys=0):
    """Predictive fs|f at the test inputs -- sliceSample.py:234-284 restated (Cholesky branch, L upper)."""
    x, y, xs = model.x, model.y, model.xs
    my = np.mean(y)
    n_samples = f.shape[1]
    ns = xs.shape[0]
    n = x.shape[0]
    m = np.tile(model.meanfunc.getMean(x), (1, n_samples))
    K = model.covfunc.getCovMatrix(x=x, mode='train')
    sn2 = model.likfunc.sn ** 2.
    L = tools.jitchol(K / sn2 + np.eye(n)).T
    alpha = tools.solve_chol(L, f - m) / sn2
    sW = np.ones((n, 1)) / np.sqrt(sn2)
    kss = model.covfunc.getCovMatrix(z=xs, mode='self_test')
    Ks = model.covfunc.getCovMatrix(x=x, z=xs, mode='cross')
    ms = model.meanfunc.getMean(xs)
    Fmu = np.tile(ms, (1, n_samples)) + np.dot(Ks.T, alpha)
    V = scipy.linalg.solve_triangular(L.T, np.tile(sW, (1, ns)) * Ks, lower=True)
    fs2 = kss - np.array([(V * V).sum(axis=0)]).T
    Fs2 = np.maximum(fs2, 0)
    Fmu = np.mean(Fmu, axis=1, keepdims=True)
    Ymu, Lower, Upper = model.likfunc.evaluate(mu=Fmu, s2=Fs2)
    ym = np.reshape(np.mean(Ymu, axis=1), (ns, 1)) + my
    ys_lw = np.reshape(np.mean(Lower, axis=1), (ns, 1)) + my
    ys_up = np.reshape(np.mean(Upper, axis=1), (ns, 1)) + my
    return ym, ys_lw, ys_up, Fs2


def elliptical_slice(f, x, y, hyp, nu=None):
    """sliceSample.py:15-74.  nu ~ N(0, K): the reference draws it with multivariate_normal (SVD of K); with
    ``nu`` given the rest is restated draw for draw (1 uniform for the height, 1 for theta, 1 per shrink)."""
    nobs = f.shape[0]
    my = np.mean(y)
    if nu is None:
        K = covK.RBF(np.log(hyp[0]), np.log(hyp[1])).getCovMatrix(x=x, mode='train')
        nu = np.random.multivariate_normal(np.zeros_like(f), K, 1).T.reshape((nobs,))
    lik = likK.TruncatedGauss2(upper=Y_UPPER - my, lower=Y_LOWER - my, log_sigma=np.log(hyp[2]))
    cur_llk = lik.evaluate(y=y - my, mu=f)
    _check_finite_start(cur_llk)
    cur_llk = cur_llk + np.log(np.random.uniform())
    theta = np.random.uniform(high=2. * np.pi)
    theta_min, theta_max = theta - 2. * np.pi, theta
    while True:
        prop_f = f * np.cos(theta) + nu * np.sin(theta)
        prop_llk = lik.evaluate(y=y - my, mu=prop_f)
        if prop_llk > cur_llk and np.isfinite(prop_llk):
            return prop_f
        if theta >= 0:
            theta_max = theta
        else:
            theta_min = theta
        theta = np.random.uniform(low=theta_min, high=theta_max)
