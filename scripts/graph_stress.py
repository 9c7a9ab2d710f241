#!/usr/bin/env python3
"""Which hipGraph object pattern is unsafe on this runtime?  (development; see gpmc_debug_graph_churn in plan.hip)

    python3 scripts/graph_stress.py FORM DEVSYNC ITERS [N [LOOKAHEAD]]

FORM 0 = the plan's segmented sequence of linear graphs, 1 = round 1's single fork/join graph; DEVSYNC 1 = a device-wide
sync between the replay and the destroys (the pattern that aborts with FORM 1 on HIP runtime 7.0.51831).

Run one mode per process: a heap corruption in the runtime aborts the interpreter.
"""
import ctypes, os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gaussianprocess-mcmc_amd'))
import gpmc
from gpmc import core, _lib

form, devsync, iters = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
mode = '%d/%d' % (form, devsync)
n = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
if len(sys.argv) > 5:
    gpmc.set_option('lookahead', int(sys.argv[5]))   # 0: single-stream launch sequence
rng = np.random.RandomState(124)
X = rng.uniform(0, 10, (n, 2)); g = rng.standard_normal(n) * 5
gpmc.set_option('use_graph', 0)          # the plan itself stays eager; only the churn entry touches graphs
plan = core.LogMarginalPlan(X, g, 'rbf')
ref = plan([1.0, 10.0, 1.2])
L = _lib.lib()
_v = ctypes.c_int(); ctypes.CDLL('libamdhip64.so.7').hipRuntimeGetVersion(ctypes.byref(_v)); print('hip runtime', _v.value, flush=True)
L.gpmc_debug_graph_churn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int]
L.gpmc_debug_graph_churn.restype = ctypes.c_int
done = 0
while done < iters:
    k = min(8, iters - done)
    _lib.check(L.gpmc_debug_graph_churn(plan._h, form, devsync, k), 'gpmc_debug_graph_churn')
    done += k
    print('form/devsync %s: %d iterations ok' % (mode, done), flush=True)
plan.close()
print('form/devsync %s: closed cleanly (log-marginal %.6f)' % (mode, ref), flush=True)
