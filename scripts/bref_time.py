#!/usr/bin/env python3
"""Time ONE literal reference-sequence SDS proposal (oracle restatement of sliceSample.py:136-147,165-207) on the host."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import sds_ref
from oracle.kcgp_ref import covK
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
rng = np.random.RandomState(124)
X = rng.uniform(0, 10, (n, 2)); g = rng.standard_normal(n) * 5; eta = rng.standard_normal(n)
hyp = np.array([1.0, 10.0, 1.2])
t0 = time.perf_counter()
K = covK.RBF(np.log(hyp[0]), np.log(hyp[1])).getCovMatrix(x=X, mode='train')
t1 = time.perf_counter()
g_, K_S, m, C, L = sds_ref.aux_var_model_literal(np.zeros(n), K, hyp[2], g)
t2 = time.perf_counter()
f = np.dot(C, eta) + m
v = sds_ref.log_marginal_literal(g, K_S, L)
t3 = time.perf_counter()
print('n=%d literal proposal: K %.2f s, aux_var_model %.2f s, f + logml %.2f s, total %.2f s  (logml %.6f)' % (n, t1 - t0, t2 - t1, t3 - t2, t3 - t0, v), flush=True)
