#!/usr/bin/env python3
"""Where does the device chain first take a different decision than the oracle (lean SciPy) chain? (development)"""
import os, sys
import numpy as np
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(R, 'gaussianprocess-mcmc_amd')); sys.path.insert(0, R)
from gpmc import framework, sliceSample
from oracle import sds_ref
n, iters = int(sys.argv[1]), int(sys.argv[2])
x, y = sds_ref.synthetic_demo_data(n=n, seed=124); y = np.clip(y, 1., 99.)
np.random.seed(5)
fc, hc, nc = sds_ref.run_chain(x, y, iters, sds_ref.DEMO_HYP0, literal=False)
print('oracle ntries', nc, flush=True)
for form in ('inv', 'syrk'):
    ws = sliceSample.SDSWorkspace(x, form=form)
    np.random.seed(5)
    fg, hg, ng = framework.run_chain(x, y, iters, framework.DEMO_HYP0, workspace=ws)
    same = ng == nc
    first = -1 if same.all() else int(np.argmin(same))
    upto = iters if first < 0 else first
    print('%-4s ntries %s  first differing iteration %d;  before it: max |hyp diff| %.2e  max |f diff| %.2e' % (
        form, ng, first, np.abs(hg[:, :upto] - hc[:, :upto]).max() if upto else 0, np.abs(fg[:, :upto] - fc[:, :upto]).max() if upto else 0), flush=True)
