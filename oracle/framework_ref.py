"""CPU restatement of the reference's cross-validation harness -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).

Follows ``/root/reference/framework.py``: ``Framework.getScaleData`` (:44-57), ``Framework.getFoldData`` (:124-147)
and ``crossValid.execute`` (:194-248), with the chain and the predictive taken from ``oracle.sds_ref``
(``run_chain`` = ``runSimulMCMC`` :59-77, ``inf_mcmc`` = sliceSample.py:234-284).

PARITY UNPINNED for this file: framework.py is Python-2 source (``print`` statements, ``/`` integer division, list
``range``) and cannot be imported here, the reference ships no test or fixture for it, and ``likK.TruncatedGauss2``
(whose ``evaluate(y=, mu=, s2=)`` produces the fold log-likelihood, :246) is missing from the repository.  What IS
pinned are the pieces it drives (chain, ``inf_mcmc``: golden fixtures).  Py2 semantics restated explicitly:
``x.shape[0]/(gap+window)`` is floor division; ``np.delete`` with the float, possibly out-of-range ``test_id`` of
NumPy 1.12 (requirements.txt:6) ignored out-of-range entries; ``range(iterMCMC*9/10-1, iterMCMC, 10)`` floors.
"""
import numpy as np

from . import sds_ref
from .kcgp_ref import covK, likK


def scale_data(y, opt):
    """framework.py:44-57: None -> unchanged, 'ctr' -> y - mean, 'std' -> (y - mean) / std (population std)."""
    y = np.asarray(y, dtype=np.float64)
    avg, stdev = np.mean(y, 0), np.std(y, 0)
    if opt == 'ctr':
        return y - avg
    if opt == 'std':
        return (y - avg) / stdev
    return y


def fold_indices(n, fold, gap, window):
    """framework.py:139-144: test ids = fold + [0, gap) + (gap + window) * i for i < n // (gap + window); the
    training set is everything else; test ids beyond the data are dropped AFTER the training set is formed."""
    test_id = []
    for i in range(n // (gap + window)):
        test_id = np.append(test_id, fold + np.arange(gap) + (gap + window) * i)
    test_id = np.asarray(test_id, dtype=np.int64)
    train_id = np.setdiff1d(np.arange(n), test_id[(test_id >= 0) & (test_id < n)], assume_unique=False)
    return train_id, test_id[test_id < n]


def get_fold_data(x, y, fold, gap, window):
    """-> train_x, train_y, test_x, test_y, test_id  (y is (n, 1); the returned targets are 1-D, framework.py:147)."""
    train_id, test_id = fold_indices(x.shape[0], fold, gap, window)
    return x[train_id, :], y[train_id, 0], x[test_id, :], y[test_id, 0], test_id


def selected_samples(iters):
    """framework.py:225."""
    return list(range(iters * 9 // 10 - 1, iters, 10))


def fold_llk(x, y, val_x, val_y, fold_f, fold_hyp, iters, pairing='literal'):
    """framework.py:221-246 for one fold: mean over the selected samples of the truncated log-likelihood of the
    predictive mean at the held-out points, per held-out point.

    pairing='literal': ``evaluate(y=ys (ns,1), mu=valY (ns,), s2=fs2 (ns,1))`` exactly as written -- NumPy broadcasts
    that to all ns x ns (prediction, target) pairs; 'matched' pairs prediction i with target i."""
    upper = 100. - np.mean(y)
    lower = 0. - np.mean(y)
    out = []
    for i in selected_samples(iters):
        ll, sf, sn = fold_hyp[0, i], fold_hyp[1, i], fold_hyp[2, i]
        prop_f = fold_f[:, i].reshape((fold_f.shape[0], 1))
        model = sds_ref.Model(x, y, val_x, covK.RBF(np.log(ll), np.log(sf)),
                              likK.TruncatedGauss2(upper=upper, lower=lower, log_sigma=np.log(sn)))
        ys, _, _, fs2 = sds_ref.inf_mcmc(prop_f, model)
        lik = model.likfunc
        lik.upper, lik.lower = 100., 0.
        if pairing == 'literal':
            out.append(lik.evaluate(y=ys, mu=val_y, s2=fs2) / ys.shape[0])
        else:
            out.append(lik.evaluate(y=ys.reshape(-1), mu=val_y, s2=fs2.reshape(-1)) / ys.shape[0])
    return float(np.mean(out))


def cross_valid(data, window, gap_array, iters, scale_opt=None, pairing='literal', seed=None, literal_chain=False):
    """crossValid.execute(updOpt='mcmcSml') -> {gap: [llk of fold 0, fold 1, ...]}.  One global legacy np.random
    stream runs through all folds in order, as in the reference (seeded once here if ``seed`` is given)."""
    x = data[:, 1:]
    y = scale_data(np.reshape(data[:, 0], (data.shape[0], 1)), scale_opt)
    if seed is not None:
        np.random.seed(seed)
    res = {}
    for gap in gap_array:
        llk = []
        for fold in range(gap + window):
            tx, ty, vx, vy, _ = get_fold_data(x, y, fold, gap, window)
            fold_f, fold_hyp, _ = sds_ref.run_chain(tx, ty, iters, (1., 10., 1.2), literal=literal_chain)
            llk.append(fold_llk(tx, ty, vx, vy, fold_f, fold_hyp, iters, pairing))
        res[gap] = llk
    return res
