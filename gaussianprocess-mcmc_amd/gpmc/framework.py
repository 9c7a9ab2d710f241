"""Chain drivers: the shape of ``Framework.runSimulMCMC`` (framework.py:59-77) and ``mcmcUpdate``
(demoRegression.py:15-32) -- same initial hyper-parameters, proposal scale, ``iter=i`` passing and history
layout (histF (n, iters), histHyp (3, iters)) -- on top of the device sampler."""
import numpy as np

from . import sliceSample

FRAMEWORK_HYP0 = (1., 10., 1.2)          # framework.py:63
DEMO_HYP0 = (0.35, 2.0, 0.2)             # demoRegression.py:15
SCALE = (10., 10., 5.)                   # framework.py:69, demoRegression.py:25


def run_chain(x, y, iters, hyp0, scale=SCALE, iter0=0, verbose=False, workspace=None):
    y = np.asarray(y, dtype=np.float64).reshape(-1)
    propHyp = np.asarray(hyp0, dtype=np.float64)
    propF = np.zeros_like(y)
    histF = np.zeros((y.shape[0], iters))
    histHyp = np.zeros((propHyp.shape[0], iters))
    ntries = np.zeros(iters, dtype=np.int64)
    ws = workspace or sliceSample.SDSWorkspace(x)
    st = {}
    for i in range(iters):
        propF, propHyp = sliceSample.surrogate_slice_sampling(propF, x, y, propHyp, scale=np.asarray(scale),
                                                              iter=iter0 + i, workspace=ws, stats=st)
        if verbose:
            print('Iteration: %r: ll=%.3f, sf=%.3f, sn=%.3f' % (i + 1, propHyp[0], propHyp[1], propHyp[2]))
        histF[:, i] = propF
        histHyp[:, i] = propHyp
        ntries[i] = st['proposals']
    return histF, histHyp, ntries


class Framework(object):
    """data: ndarray (n, 1 + d) = [target | inputs] as at framework.py:24-28."""

    def __init__(self, data):
        if data is None:
            raise Exception('No data is given.')
        self.x = data[:, 1:]
        self.y = np.reshape(data[:, 0], (np.shape(data)[0], 1))

    def runSimulMCMC(self, iters, verbose=False):
        histF, histHyp, _ = run_chain(self.x, self.y.reshape(-1), iters, FRAMEWORK_HYP0, verbose=verbose)
        return histF, histHyp


def mcmcUpdate(data, n_samples, ll=0.35, sf=2.0, sn=0.2, verbose=False):
    """demoRegression.py:15-32: returns (histHyp, histF)."""
    histF, histHyp, _ = run_chain(data['x'], data['y'], n_samples, (ll, sf, sn), verbose=verbose)
    return histHyp, histF


def synthetic_demo_data(n=455, seed=124, ll=5.0, sf=20.0, sn=2.5, mean=91.1538461538):
    """demoRegression.py:116-131 on the device: x = arange(n), y = chol(K + sn^2 I) z + mean."""
    import torch
    from . import covK, ops, tools
    np.random.seed(seed)
    x = np.arange(0, n).astype(np.float64).reshape(n, 1)
    K = covK.RBF(np.log(ll), np.log(sf)).getCovMatrix(x, mode='train', diag_add=sn ** 2)
    L = tools.jitchol(K)
    z = ops.to_dev(np.random.normal(size=(n,)))
    y = torch.mv(L, z).cpu().numpy() + mean      # one O(N^2) product to draw the data, not on the hot path
    return x, y


def output(x, y, gap=0, histHyp=None, histF=None, llk=None, outdir='./output'):
    """CSV files in the reference's layout (Framework.output, framework.py:79-122): ``hypGap<gap>.csv`` with header
    ll,sf2,sn and one row per hyper-parameter, ``fGap<gap>.csv`` with columns 1..iters,x,y, ``llkGap<gap>.csv``."""
    import csv
    import os
    os.makedirs(outdir, exist_ok=True)
    if histHyp is not None:
        with open(os.path.join(outdir, 'hypGap%s.csv' % gap), 'w', newline='') as h:
            w = csv.writer(h)
            w.writerow(['ll', 'sf2', 'sn'])
            w.writerows(histHyp)
    if histF is not None:
        with open(os.path.join(outdir, 'fGap%s.csv' % gap), 'w', newline='') as fh:
            w = csv.writer(fh)
            w.writerow(list(range(1, histF.shape[1] + 1)) + ['x', 'y'])
            xy = np.hstack((np.asarray(x).reshape(-1, 1), np.asarray(y).reshape(-1, 1)))
            w.writerows(np.hstack((histF, xy)))
    if llk is not None:
        with open(os.path.join(outdir, 'llkGap%s.csv' % gap), 'w', newline='') as k:
            w = csv.writer(k)
            w.writerow(['gap'] + [str(i) for i in range(len(llk))])
            w.writerow([gap] + list(llk))
    return 0


def outputDemoRes(x, y, histF, histHyp, outdir='./output'):
    """demoRegression.py:51-73: demo_f.csv (columns 1..n,x,y) and demo_hyp.csv (header ll,sf2,sn, one row per sample)."""
    import csv
    import os
    os.makedirs(outdir, exist_ok=True)
    with open(os.path.join(outdir, 'demo_f.csv'), 'w', newline='') as fh:
        w = csv.writer(fh)
        w.writerow(list(range(1, histF.shape[1] + 1)) + ['x', 'y'])
        w.writerows(np.hstack((histF, np.hstack((np.asarray(x).reshape(-1, 1), np.asarray(y).reshape(-1, 1))))))
    with open(os.path.join(outdir, 'demo_hyp.csv'), 'w', newline='') as h:
        w = csv.writer(h)
        w.writerow(['ll', 'sf2', 'sn'])
        w.writerows(histHyp.T)
