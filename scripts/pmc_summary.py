#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; one counter per pass) per kernel.

    python3 scripts/pmc_summary.py OUT.json DIR_OR_CSV [DIR_OR_CSV ...]

Reads every *counter_collection.csv below the given paths.  Units and corrections follow
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): both counters are in KB; on gfx950 FETCH_SIZE reports one
half of the bytes of wide coalesced reads, so it is doubled; WRITE_SIZE is exact.  Output: per kernel the number
of dispatches and the mean HBM-side bytes per launch (fetch corrected, write, and their sum).
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(.*$', '', name)                # drop the argument list
    name = re.sub(r'^void\s+', '', name)
    return name.strip()


def main():
    out, paths = sys.argv[1], sys.argv[2:]
    files = []
    for p in paths:
        if os.path.isdir(p):
            files += glob.glob(os.path.join(p, '**', '*counter_collection.csv'), recursive=True)
        else:
            files.append(p)
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))      # kernel -> counter -> [dispatches, sum]
    for f in sorted(files):
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                k = short(row['Kernel_Name'])
                a = acc[k][row['Counter_Name']]
                a[0] += 1
                a[1] += float(row['Counter_Value'])
    res = {}
    for k, cs in acc.items():
        e = {}
        if 'FETCH_SIZE' in cs:
            n, s = cs['FETCH_SIZE']
            e['launches_fetch_pass'] = n
            e['fetch_bytes_per_launch'] = 2.0 * 1024.0 * s / n      # KB -> B, x2 gfx950 correction
        if 'WRITE_SIZE' in cs:
            n, s = cs['WRITE_SIZE']
            e['launches_write_pass'] = n
            e['write_bytes_per_launch'] = 1024.0 * s / n
        if 'fetch_bytes_per_launch' in e and 'write_bytes_per_launch' in e:
            e['hbm_bytes_per_launch'] = e['fetch_bytes_per_launch'] + e['write_bytes_per_launch']
        res[k] = e
    doc = {'source': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes; FETCH_SIZE x2 (gfx950), KB -> B',
           'files': [os.path.relpath(f) for f in sorted(files)], 'kernels': res}
    with open(out, 'w') as fh:
        json.dump(doc, fh, indent=1, sort_keys=True)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1].get('hbm_bytes_per_launch', 0)):
        print('%-70s %s' % (k[:70], {a: (round(b / 1e6, 2) if 'bytes' in a else b) for a, b in e.items()}))


if __name__ == '__main__':
    main()
