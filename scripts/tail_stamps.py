#!/usr/bin/env python3
"""Where the tile-DAG kernel's critical chain spends its time: s_memtime stamps of the pair workgroups (development).

    python3 scripts/tail_stamps.py [N] [name=value ...]
"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gaussianprocess-mcmc_amd'))
import gpmc
from gpmc import core, _lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
gpmc.set_option('use_graph', 0)
for a in sys.argv[2:]:
    k, v = a.split('=')
    gpmc.set_option(k, int(v))
rng = np.random.RandomState(124)
X = rng.uniform(0, 10, (n, 2)); g = rng.standard_normal(n) * 5
plan = core.LogMarginalPlan(X, g, 'rbf')
plan([1.0, 10.0, 1.2])
dbg = torch.zeros(16 * 64, dtype=torch.int64, device='cuda')
L = _lib.lib()
L.gpmc_debug_tail_stamps.argtypes = [ctypes.c_void_p]
L.gpmc_debug_tail_stamps(dbg.data_ptr())
plan([1.0, 10.0, 1.2])
torch.cuda.synchronize()
L.gpmc_debug_tail_stamps(None)
s = dbg.cpu().numpy().reshape(64, 16).astype(np.float64)
T = ops_T = (gpmc.ops.pad(n)) // 64
names = ['start->accum done', 'wait F[l-1]', 'load L + inverses', 'TRSM', 'stage+store issue+D update', 'drain+flag', 'factor', 'store+drain+flag']
print('n=%d T=%d; ticks are s_memtime units' % (n, T))
rows = range(4, T)
tot = []
for l in rows:
    d = np.diff(s[l, :9])
    tot.append(d)
tot = np.array(tot)
for k, nm in enumerate(names):
    print('%-30s mean %8.0f  min %8.0f  max %8.0f' % (nm, tot[:, k].mean(), tot[:, k].min(), tot[:, k].max()))
print('   of "stage+store issue+D update": staging+barrier %.0f, store issue %.0f, D update %.0f' % (
    (s[4:T, 9] - s[4:T, 4]).mean(), (s[4:T, 10] - s[4:T, 9]).mean(), (s[4:T, 5] - s[4:T, 10]).mean()))
# (stamps of different workgroups are not comparable: s_memtime counts per XCD)
plan.close()
