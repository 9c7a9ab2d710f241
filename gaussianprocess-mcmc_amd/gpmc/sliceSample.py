"""``kcMCMC.sliceSample`` mirror: surrogate-data slice sampling with the dense algebra on the MI355X.

``surrogate_slice_sampling(f, x, y, hyp, scale, iter=0)`` keeps the reference's signature, return value,
hard-coded constants and -- draw for draw -- its use of the global legacy ``np.random`` stream
(kcMCMC/sliceSample.py:76-163), so a seeded chain is comparable with the CPU oracle.
Control flow stays on the host exactly as in the reference; per proposal one scalar (the log-marginal) and one
N-vector (the proposed latent f', needed by the O(N) likelihood) cross back.
"""
import numpy as np
import scipy.special
import torch

from . import likK, ops
from .sds import SDSWorkspace, aux_var_model  # noqa: F401  (aux_var_model re-exported: same module layout)

PRIOR_K = np.asarray([1., 3., 3.])          # sliceSample.py:124
PRIOR_THETA = np.asarray([1., 1.5, 3.])     # sliceSample.py:125
NOISE_FREEZE_ITERS = 500                    # sliceSample.py:128,133,151
Y_UPPER, Y_LOWER = 100., 0.                 # sliceSample.py:114-115

_ws_cache = {}


def log_gamma(x, k, theta, invG):
    """Gamma / inverse-Gamma log-pdf and gradient (sliceSample.py:209-232); O(1) host arithmetic."""
    x = np.asarray(x, dtype=np.float64)
    logG = (k - 1) * np.log(x) - x / theta - k * np.log(theta) - np.log(scipy.special.gamma(k))
    gradG = (k - 1) * (1 / x) - 1 / theta
    if invG:
        logG[2] = (np.log(theta[2] ** k[2]) - np.log(scipy.special.gamma(k[2])) + (-k[2] - 1) * np.log(x[2])
                   + (-theta[2] / x[2]))
        gradG[2] = (-k[2] - 1) / x[2] + theta[2] / (x[2] ** 2)
    return logG, gradG


def workspace_for(x):
    """One SDSWorkspace per training-input array (keyed by identity and shape), so the chain drivers'
    repeated calls reuse ~3 N^2 of device memory instead of reallocating it."""
    key = (id(x), tuple(np.shape(x)))
    ws = _ws_cache.get(key)
    if ws is None:
        _ws_cache.clear()
        ws = _ws_cache[key] = SDSWorkspace(x)
    return ws


def surrogate_slice_sampling(f, x, y, hyp, scale, iter=0, workspace=None, stats=None):
    ws = workspace or workspace_for(x)
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    hyp = np.asarray(hyp, dtype=np.float64)
    scale = np.asarray(scale, dtype=np.float64)
    n = ws.n
    my = np.mean(y)

    f_host = np.asarray(f.detach().cpu().numpy() if isinstance(f, torch.Tensor) else f, dtype=np.float64).reshape(-1)
    f_dev = ops.to_dev(f_host)
    # g ~ N(f, S): N standard normals from the legacy stream (bit-identical to the reference's
    # multivariate_normal(f, S, 1) for S = s I, see oracle/sds_ref.draw_g and tests/golden/gdraw_*.npz)
    from . import covK
    Sii = covK.s_diag_scalar(np.exp(2. * np.log(hyp[1])), hyp[2])
    g_host = f_host + np.sqrt(Sii) * np.random.standard_normal(n)
    g = ops.to_dev(g_host)

    curG = ws.aux_var_model(f_dev, hyp, g, need_eta=True)
    ita = ws.eta.clone()

    v = np.random.uniform(low=0., high=scale)
    hyp_min = np.maximum(hyp - v, 0)
    hyp_max = hyp_min + scale

    lik_func = likK.TruncatedGauss2(upper=Y_UPPER - my, lower=Y_LOWER - my, log_sigma=np.log(hyp[2]))
    cur_llk = lik_func.evaluate(y=y - my, mu=f_host)

    prior, _ = log_gamma(hyp, PRIOR_K, PRIOR_THETA, True)
    threshold = np.log(np.random.uniform()) + cur_llk + prior[1] + prior[0] + curG
    if iter >= NOISE_FREEZE_ITERS:
        threshold += prior[2]

    ntry = 0
    while True:
        ntry += 1
        prop_hyp = np.random.uniform(low=hyp_min, high=hyp_max)
        if iter < NOISE_FREEZE_ITERS:
            prop_hyp[2] = hyp[2]

        propG = ws.aux_var_model(f_dev, prop_hyp, g, need_eta=False)   # LinAlgError propagates, as in the reference
        prop_f = ws.propose_f(ita).cpu().numpy()
        lik_func.sn = prop_hyp[2]
        prop_llk = lik_func.evaluate(y=y - my, mu=prop_f)
        propPrior, _ = log_gamma(prop_hyp, PRIOR_K, PRIOR_THETA, True)
        proposal = prop_llk + propPrior[1] + propPrior[0] + propG
        if iter >= NOISE_FREEZE_ITERS:
            proposal += propPrior[2]

        if proposal > threshold and np.isfinite(proposal):
            if stats is not None:
                stats['proposals'] = ntry
            return prop_f, prop_hyp
        for i in range(0, 3):
            if prop_hyp[i] < hyp[i]:
                hyp_min[i] = prop_hyp[i]
            else:
                hyp_max[i] = prop_hyp[i]
