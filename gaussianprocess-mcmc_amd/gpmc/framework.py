"""Chain drivers: the shape of ``Framework.runSimulMCMC`` (framework.py:59-77) and ``mcmcUpdate``
(demoRegression.py:15-32) -- same initial hyper-parameters, proposal scale, ``iter=i`` passing and history
layout (histF (n, iters), histHyp (3, iters)) -- on top of the device sampler."""
import numpy as np

from . import sliceSample

FRAMEWORK_HYP0 = (1., 10., 1.2)          # framework.py:63
DEMO_HYP0 = (0.35, 2.0, 0.2)             # demoRegression.py:15
SCALE = (10., 10., 5.)                   # framework.py:69, demoRegression.py:25


def run_chain(x, y, iters, hyp0, scale=SCALE, iter0=0, verbose=False, workspace=None, margins=None):
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
        if margins is not None:             # per iteration: proposal - threshold of every try
            margins.append(list(st['margins']))
    return histF, histHyp, ntries


class Framework(object):
    """data: ndarray (n, 1 + d) = [target | inputs] as at framework.py:24-28; ``window`` / ``gap`` are accepted and
    ignored by the base class exactly as there."""

    def __init__(self, data, window=None, gap=None):
        if data is None:
            raise Exception('No data is given.')
        self.x = data[:, 1:]
        self.y = np.reshape(data[:, 0], (np.shape(data)[0], 1))
        self.verbose = False
        self.setScaleOpt(None)

    def execute(self):
        pass

    def setScaleOpt(self, scale):
        """framework.py:39-41 (the reference prints the option; here only when ``self.verbose``)."""
        self.scaleOpt = scale
        if self.verbose:
            print('Current data scaling option: %s' % (self.scaleOpt))

    def getScaleData(self):
        """framework.py:43-57: 'ctr' -> y - mean, 'std' -> (y - mean) / std (population std), else unchanged."""
        avg = np.mean(self.y, 0)
        stdev = np.std(self.y, 0)
        if self.scaleOpt == 'ctr':
            self.y = self.y - avg
        elif self.scaleOpt == 'std':
            self.y = (self.y - avg) / stdev

    def runSimulMCMC(self, iters, verbose=False):
        histF, histHyp, _ = run_chain(self.x, np.asarray(self.y).reshape(-1), iters, FRAMEWORK_HYP0, verbose=verbose)
        return histF, histHyp

    def output(self, gap=0, histHyp=None, histF=None, llk=None, outdir='./output'):
        """framework.py:79-122 (``fGap`` assumes 1-D inputs like the reference: x is reshaped to one column)."""
        return output(self.x, self.y, gap, histHyp, histF, llk, outdir)

    def getFoldData(self, fold, gap, window):
        """framework.py:124-147 -> train_x, train_y, test_x, test_y, test_id.

        Test ids are ``fold + [0, gap) + (gap + window) * i`` for ``i < n // (gap + window)`` (Python-2 ``/`` on ints);
        everything else trains; ids past the end of the data are dropped after the training set is formed (the
        NumPy 1.12 ``np.delete`` of the reference ignored them)."""
        n = self.x.shape[0]
        test_id = []
        for i in range(n // (gap + window)):
            test_id = np.append(test_id, fold + np.arange(gap) + (gap + window) * i)
        test_id = np.asarray(test_id, dtype=np.int64)
        keep = np.ones(n, dtype=bool)
        keep[test_id[(test_id >= 0) & (test_id < n)]] = False
        train_id = np.nonzero(keep)[0]
        test_id = test_id[test_id < n]
        return self.x[train_id, :], self.y[train_id, 0], self.x[test_id, :], self.y[test_id, 0], test_id


class singleRun(Framework):
    """First framework: single-run GP (framework.py:149-175).  ``updOpt='mcmcSml'`` runs the chain and writes the CSVs;
    ``'opt'`` is kcGP.gpK.GPR.optimize, which is out of scope."""

    def __init__(self, data):
        super(singleRun, self).__init__(data, window=None, gap=None)
        self.name = 'First framework: single-run GP'

    def execute(self, updOpt=None, iterMCMC=1000, outdir='./output'):
        self.getScaleData()
        assert updOpt is not None, 'Please choose either optimization or MCMC'
        if updOpt == 'opt':
            raise NotImplementedError("updOpt='opt' (GPR.optimize) is outside the MCMC path this package replaces")
        if updOpt == 'mcmcSml':
            histF, histHyp = self.runSimulMCMC(iterMCMC)
            self.output(histHyp=histHyp.T, histF=histF, outdir=outdir)
            return histF, histHyp


def fold_owner(task, world):
    """(gap, fold) task number -> rank: round robin, the chain-per-GPU launcher of SURVEY.md 8(f) rank 4."""
    return task % world


def selected_samples(iters):
    """framework.py:225 (Python-2 integer division)."""
    return list(range(iters * 9 // 10 - 1, iters, 10))


class crossValid(Framework):
    """Second framework: GP with cross validation over (gap, fold) (framework.py:177-248).

    Every fold is an independent chain plus a handful of ``inf_mcmc`` predictives, so folds are the shard: with a
    process group of ``world`` ranks (one per GPU) task t = (gap, fold) in enumeration order runs on rank
    ``t % world``; ONE all_gather of the per-fold log-likelihoods ends ``execute`` (no data-path collective).

    Random stream: a single process consumes the global legacy ``np.random`` stream fold after fold exactly like the
    reference.  With more than one rank (or ``seed_base`` given) fold t is seeded ``seed_base + t`` so the result does
    not depend on the partition.

    ``pairing='literal'`` evaluates ``trunclik.evaluate(y=ys, mu=valY, s2=fs2)/ys.shape[0]`` as written at
    framework.py:246 -- ys is (ns, 1) and valY (ns,), which NumPy broadcasts to all ns x ns pairs; ``'matched'`` pairs
    prediction i with target i.  TruncatedGauss2 is inferred (likK.py), so these numbers are parity-unpinned.
    """

    def __init__(self, data, window, gapArray):
        super(crossValid, self).__init__(data, window, gapArray)
        self.name = 'Second framework: GP with window width: %d' % (window)
        self.windowSize = window
        self.gapArray = gapArray
        self.testLLK = {}

    def foldLLK(self, x, y, valX, valY, foldF, foldHyp, iterMCMC, pairing='literal'):
        """framework.py:221-246 for one fold (x, y = the fold's training data)."""
        from . import covK, gpK, likK
        model = gpK.GPR()
        model.xs = valX
        upper = 100. - np.mean(y)
        lower = 0. - np.mean(y)
        out = []
        for i in selected_samples(iterMCMC):
            ll, sf, sn = foldHyp[0, i], foldHyp[1, i], foldHyp[2, i]
            propF = foldF[:, i].reshape((foldF.shape[0], 1))
            model.setPrior(kernel=covK.RBF(np.log(ll), np.log(sf)))
            model.setNoise(np.log(sn))
            model.setData(x, y)
            trunclik = likK.TruncatedGauss2(upper=upper, lower=lower, log_sigma=np.log(sn))
            model.likfunc = trunclik
            ys, _, _, fs2 = sliceSample.inf_mcmc(propF, model)          # inference fs|f
            trunclik.upper = 100.
            trunclik.lower = 0.
            if pairing == 'literal':
                out.append(trunclik.evaluate(y=ys, mu=valY, s2=fs2) / ys.shape[0])
            else:
                out.append(trunclik.evaluate(y=ys.reshape(-1), mu=valY, s2=np.asarray(fs2).reshape(-1)) / ys.shape[0])
        return float(np.mean(out))

    def runFold(self, gap, fold, iterMCMC, pairing='literal'):
        """One (gap, fold) task on this process's GPU -> (llk, foldF, foldHyp); self.x / self.y are left on the
        fold's training data like the reference does."""
        self.x, self.y = self._originalX, self._originalY
        self.x, self.y, valX, valY, _ = self.getFoldData(fold, gap, window=self.windowSize)
        foldF, foldHyp = self.runSimulMCMC(iterMCMC)
        return self.foldLLK(self.x, self.y, valX, valY, foldF, foldHyp, iterMCMC, pairing), foldF, foldHyp

    def execute(self, updOpt=None, iterMCMC=1000, pairing='literal', seed_base=None, outdir='./output', fold_fn=None):
        """-> {gap: [llk per fold]} on every rank (also kept in ``self.testLLK``); rank 0 writes the reference's
        CSVs per gap (hyper-parameter / latent histories of the last fold IT ran, like the reference's last fold).
        ``fold_fn(gap, fold, iterMCMC) -> (llk, foldF, foldHyp)`` replaces ``runFold`` (used by the CPU tests of the
        fan-out)."""
        import torch.distributed as dist
        assert updOpt is not None, 'Please choose either optimization or MCMC'
        if updOpt == 'opt':
            raise NotImplementedError("updOpt='opt' (GPR.optimize) is outside the MCMC path this package replaces")
        self.getScaleData()
        self._originalX, self._originalY = self.x[:], self.y[:]
        rank, world = (dist.get_rank(), dist.get_world_size()) if (dist.is_available() and dist.is_initialized()) else (0, 1)
        if world > 1 and seed_base is None:
            seed_base = 124
        run = fold_fn or (lambda g, f, it: self.runFold(g, f, it, pairing))
        tasks = [(gap, fold) for gap in self.gapArray for fold in range(gap + self.windowSize)]
        llk = np.full(len(tasks), np.nan)
        last = {}
        for t, (gap, fold) in enumerate(tasks):
            if fold_owner(t, world) != rank:
                continue
            if seed_base is not None:
                np.random.seed(seed_base + t)
            if self.verbose:
                print('Gap: %r, fold %d' % (gap, fold + 1))
            llk[t], foldF, foldHyp = run(gap, fold, iterMCMC)
            last[gap] = (foldF, foldHyp, self.x, self.y)      # the reference writes the gap's last fold with ITS x, y
        if world > 1:
            from . import chains
            llk = chains.gather_vector(llk)           # the one collective: every rank gets every fold's number
        res, t = {}, 0
        for gap in self.gapArray:
            nf = gap + self.windowSize
            res[gap] = [float(v) for v in llk[t:t + nf]]
            t += nf
            if rank == 0 and outdir is not None and gap in last:
                foldF, foldHyp, fx, fy = last[gap]
                output(fx, fy, gap, foldHyp.T, foldF, res[gap], outdir)
        self.testLLK = res
        return res


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
    L = ops.to_dev(tools.jitchol(K))             # (a tensor whatever gpmc.set_return says)
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
