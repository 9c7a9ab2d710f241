#!/usr/bin/env python3
"""Diagnostic: s_memtime phase stamps of the fused panel kernel (DBG instantiation) on a random SPD workspace."""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'gaussianprocess-mcmc_amd'))
from gpmc import _lib, ops, tools
n = 8192
ws = tools.FactorWorkspace(n, 1)
ws.W.normal_(); ws.W.diagonal()[:n].add_(1e4)
lib = _lib.lib()
fn = lib.gpmc_debug_panel_stamps
fn.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
               ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]
dbg = torch.zeros(64, dtype=torch.int64, device='cuda')
names = ['start->loads issued', 'K loop', 'D/T form (incl. strip LDS transpose)', 'factor (mp_factor64)', 'to inverses', 'inverses',
         'TRSM', 'store']
for k0, c0 in ((0, 0), (0, 64), (0, 128), (0, 192)):
    for rep in range(3):
        rc = fn(ws.W.data_ptr(), ws.W.stride(0), ws.mp, k0, c0, ws.info.data_ptr(), ws.scratch.data_ptr(), ws.np_,
                dbg.data_ptr(), ops.stream_ptr())
        assert rc == 0, lib.gpmc_last_error()
        torch.cuda.synchronize()
    tall = dbg.cpu().numpy().astype(np.int64)
    t = tall[:8]
    d = np.diff(t)
    print('K=%3d total %6d ticks: ' % (c0 - k0, t[7] - t[0]) + '  '.join('%s=%d' % (nm.split(' ')[0], x) for nm, x in zip(names[:7], d)))
    if c0 == 0:
        for w in range(5):
            tw = tall[8 + 8 * w: 8 + 8 * w + 5]
            print('   micro-step 3, wave %d: pre-B1 work=%d  wait B1=%d  factor/shadow work=%d  wait B2=%d   (start %d)' % ((w,) + tuple(np.diff(tw)) + (tw[0] - tall[8],)))
print('(s_memtime ticks; 100 MHz reference => 1 tick = 10 ns if constant-rate, else shader cycles)')
