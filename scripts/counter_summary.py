#!/usr/bin/env python3
"""Per-kernel means of every counter in rocprofv3 --pmc passes (one row per dispatch and counter).

    python3 scripts/counter_summary.py OUT.json DIR [DIR ...]

For each kernel: dispatches seen, and per counter the mean value per dispatch (a counter collected in several passes is
averaged over all of them).  Derived, where the inputs are there (MI355X_MICROARCH.md: SQ_WAVE_CYCLES / SQ_WAIT_* /
SQ_ACTIVE_INST_* count quad-cycles, SQ_VALU_MFMA_BUSY_CYCLES and GRBM_GUI_ACTIVE count cycles; GRBM_GUI_ACTIVE is the sum
over the 8 XCDs; SQ_BUSY_CYCLES is per shader engine):
  kernel_cycles      = GRBM_GUI_ACTIVE / 8          (cycles the launch was resident, per XCD)
  mfma_busy_pct      = SQ_VALU_MFMA_BUSY_CYCLES / (kernel_cycles * 1024 SIMDs)
  mfma_f64_tflop     = SQ_INSTS_VALU_MFMA_MOPS_F64 * 512 flop
  wait_inst_pct      = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES     (issue stalls: MFMA pipe / dependencies)
  wait_any_pct       = SQ_WAIT_ANY / SQ_WAVE_CYCLES          (parked at s_waitcnt / barriers / sleeps)
  lds_busy_pct       = SQ_LDS_IDX_ACTIVE / (kernel_cycles * 256 CUs)
  lds_conflict_pct   = SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE
  l2_hit_pct         = TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum)
"""
import csv, glob, json, os, re, sys
from collections import defaultdict


def short(name):
    name = re.sub(r'\(.*$', '', name)
    name = re.sub(r'^void\s+', '', name)
    return name.strip()


def main():
    out, paths = sys.argv[1], sys.argv[2:]
    acc = defaultdict(lambda: defaultdict(lambda: [0, 0.0]))
    for p in paths:
        for f in sorted(glob.glob(os.path.join(p, '**', '*counter_collection.csv'), recursive=True)):
            with open(f, newline='') as fh:
                for row in csv.DictReader(fh):
                    a = acc[short(row['Kernel_Name'])][row['Counter_Name']]
                    a[0] += 1
                    a[1] += float(row['Counter_Value'])
    res = {}
    for k, cs in acc.items():
        e = {'dispatches': max(v[0] for v in cs.values())}
        m = {c: v[1] / v[0] for c, v in cs.items()}
        e['mean_per_dispatch'] = m
        d = {}
        if 'GRBM_GUI_ACTIVE' in m:
            kc = m['GRBM_GUI_ACTIVE'] / 8.0
            d['kernel_cycles'] = kc
            if 'SQ_VALU_MFMA_BUSY_CYCLES' in m:
                d['mfma_busy_pct'] = 100.0 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (kc * 1024.0)
            if 'SQ_LDS_IDX_ACTIVE' in m:
                d['lds_busy_pct'] = 100.0 * m['SQ_LDS_IDX_ACTIVE'] / (kc * 256.0)
        if 'SQ_INSTS_VALU_MFMA_MOPS_F64' in m:
            d['mfma_f64_gflop_per_dispatch'] = m['SQ_INSTS_VALU_MFMA_MOPS_F64'] * 512.0 / 1e9
        if m.get('SQ_WAVE_CYCLES'):
            for c, nm in (('SQ_WAIT_INST_ANY', 'wait_inst_pct'), ('SQ_WAIT_ANY', 'wait_any_pct'), ('SQ_ACTIVE_INST_ANY', 'active_inst_pct')):
                if c in m:
                    d[nm] = 100.0 * m[c] / m['SQ_WAVE_CYCLES']
        if m.get('SQ_LDS_IDX_ACTIVE') and 'SQ_LDS_BANK_CONFLICT' in m:
            d['lds_conflict_pct'] = 100.0 * m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']
        if 'TCC_HIT_sum' in m and 'TCC_MISS_sum' in m and m['TCC_HIT_sum'] + m['TCC_MISS_sum'] > 0:
            d['l2_hit_pct'] = 100.0 * m['TCC_HIT_sum'] / (m['TCC_HIT_sum'] + m['TCC_MISS_sum'])
        e['derived'] = d
        res[k] = e
    with open(out, 'w') as fh:
        json.dump({'source': 'rocprofv3 --pmc passes, means per dispatch', 'kernels': res}, fh, indent=1, sort_keys=True)
    for k, e in sorted(res.items(), key=lambda kv: -kv[1]['mean_per_dispatch'].get('GRBM_GUI_ACTIVE', 0)):
        print('%s  (%d dispatches)' % (k[:100], e['dispatches']))
        print('   derived: ' + ', '.join('%s=%.4g' % kv for kv in sorted(e['derived'].items())))
        print('   raw/1e6: ' + ', '.join('%s=%.4g' % (c, v / 1e6) for c, v in sorted(e['mean_per_dispatch'].items())))


if __name__ == '__main__':
    main()
