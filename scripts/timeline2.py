#!/usr/bin/env python3
"""Per-evaluation kernel timeline from a rocprofv3 kernel_trace.csv: every kernel of the LAST evaluation with start,
duration, gap to the previous end on its queue (development aid).   python3 scripts/timeline2.py DIR [max rows]"""
import csv, glob, sys
d = sys.argv[1]
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 80
f = glob.glob(d + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# an evaluation ends with its reduction (a plan may split its K build in two launches, or have none: tile generator)
idx = [0] + [i + 1 for i, r in enumerate(rows) if 'logdet' in r['Kernel_Name'] and i + 1 < len(rows)]
segs = [rows[a:b] for a, b in zip(idx, idx[1:] + [len(rows)])]
def name(r):
    k = r['Kernel_Name']
    for key, nm in (('tail_kernel', 'DAG'), ('panel_block', 'panel'), ('panel_pair', 'pair'), ('logdet', 'logdet'), ('kbuild', 'kbuild'), ('flush', 'flush')):
        if key in k: return nm
    if 'gemm4' in k: return 'gemmBIG' if '16, 4' in k else 'gemmsm'
    return k.replace('void gpmc::', '')[:16]
seg = segs[-2] if len(segs) > 1 else segs[-1]
t0 = int(seg[0]['Start_Timestamp'])
print('evals: %d; showing one with %d kernels, span %.3f ms' % (len(segs), len(seg), (max(int(r['End_Timestamp']) for r in seg) - t0) / 1e6))
tot = {}
for r in seg:
    v = tot.setdefault(name(r), [0, 0]); v[0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); v[1] += 1
print('   ' + '  '.join('%s: %.2f ms (%d)' % (k, v[0] / 1e6, v[1]) for k, v in tot.items()))
lastend = {}
for r in seg[:nshow]:
    q = r['Queue_Id']; s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    gap = (s - lastend[q]) / 1e3 if q in lastend else 0.0
    lastend[q] = e
    print('   %-8s q=%-3s start=%8.1f dur=%7.1f end=%8.1f gap_on_queue=%6.1f wgs=%5d' % (name(r), q, (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, gap, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X'])))
