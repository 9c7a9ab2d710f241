#!/usr/bin/env python3
"""Kernel timeline of one SDS proposal from a rocprofv3 kernel trace of scripts/sds_time.py (development).
    python3 scripts/sds_timeline.py DIR"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# a proposal ends with publish_host
ends = [i for i, r in enumerate(rows) if 'publish_host' in r['Kernel_Name']]
segs = [rows[a + 1:b + 1] for a, b in zip(ends, ends[1:])]
segs = [s for s in segs if 20 <= len(s) <= 40]
seg = sorted(segs, key=lambda s: int(s[-1]['End_Timestamp']) - int(s[0]['Start_Timestamp']))[len(segs) // 2]
t0 = int(seg[0]['Start_Timestamp'])
print('proposals: %d; the median one: %d kernels, span %.3f ms' % (len(segs), len(seg), (int(seg[-1]['End_Timestamp']) - t0) / 1e6))
last = {}
busy_until = t0
idle = 0.0
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s > busy_until: idle += (s - busy_until) / 1e3
    busy_until = max(busy_until, e)
    nm = r['Kernel_Name'].replace('void ', '').replace('gpmc::', '')[:44]
    print('  %-44s q=%-2s start=%9.1f dur=%8.1f end=%9.1f wgs=%d' % (nm, r['Queue_Id'], (s - t0) / 1e3, (e - s) / 1e3, (e - t0) / 1e3, int(r['Grid_Size_X']) // max(1, int(r['Workgroup_Size_X']))))
print('idle (no kernel running) inside the proposal: %.1f us' % idle)
