#!/usr/bin/env python3
"""Print per-evaluation kernel timelines from a rocprofv3 kernel_trace.csv (development aid)."""
import csv, glob, sys
d = sys.argv[1]
nshow = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = glob.glob(d + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'kbuild' in r['Kernel_Name']]
segs = [rows[a:b] for a, b in zip(idx, idx[1:] + [len(rows)])]
def name(r):
    k = r['Kernel_Name']
    if 'panel_block' in k: return 'panel'
    if '4, 4, false' in k or '4, 4, true' in k: return 'gemmBIG'
    if '2, 2, false' in k or '2, 2, true' in k: return 'gemmsm'
    if 'gemm4' in k: return 'gemm4'
    return k.replace('void gpmc::', '')[:18]
sizes = sorted(set(len(s) for s in segs))
print('evals:', len(segs), 'kernel counts:', sizes)
for want in sizes:
    cand = [s for s in segs if len(s) == want]
    for tag, seg in (('first', cand[0]), ('last', cand[-1])):
        t0 = int(seg[0]['Start_Timestamp'])
        print('== %s eval with %d kernels: span %.3f ms' % (tag, want, (max(int(r['End_Timestamp']) for r in seg) - t0) / 1e6))
        tot = {}
        for r in seg:
            tot.setdefault(name(r), [0, 0]); tot[name(r)][0] += int(r['End_Timestamp']) - int(r['Start_Timestamp']); tot[name(r)][1] += 1
        print('   ' + '  '.join('%s: %.2f ms (%d)' % (k, v[0] / 1e6, v[1]) for k, v in tot.items()))
        if tag == 'last':
            for r in seg[:nshow]:
                print('   %-8s start=%8.1f dur=%7.1f wgs=%5d q=%s' % (name(r), (int(r['Start_Timestamp']) - t0) / 1e3,
                      (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3, int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), r['Queue_Id']))
