#!/usr/bin/env python3
"""Per-SIMD summary of a k1_anatomy wave dump: usage wave_csv.py file.csv"""
import csv, sys, collections
import numpy as np
for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    st = np.array([float(r['start_us']) for r in rows]); en = np.array([float(r['end_us']) for r in rows])
    cy = np.array([int(r['cycles']) for r in rows]); nr = np.array([int(r['rounds']) for r in rows])
    hw = np.array([int(r['hw_id']) for r in rows]); xcc = np.array([int(r['xcc']) for r in rows])
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = xcc * 100000 + se * 1000 + sh * 500 + cu * 10 + simd
    d = collections.defaultdict(list)
    for i, k in enumerate(key): d[k].append(i)
    print(f, len(rows), "waves; SIMDs", len(d), "waves/SIMD", dict(collections.Counter(len(v) for v in d.values())))
    print("  wave start pct 0/50/100", np.percentile(st, [0, 50, 100]), " end pct 0/10/50/90/100", np.percentile(en, [0, 10, 50, 90, 100]))
    tot = np.array([nr[v].sum() for v in d.values()]); endm = np.array([en[v].max() for v in d.values()])
    for t in sorted(set(tot)):
        m = tot == t
        print("  SIMDs with %d rounds: %d, last wave ends mean %.2f min %.2f max %.2f us" % (t, m.sum(), endm[m].mean(), endm[m].min(), endm[m].max()))
    print("  rounds per wave:", dict(collections.Counter(nr)), " life by rounds:", {int(k): round(float(np.mean((en - st)[nr == k])), 2) for k in sorted(set(nr))})
    print("  clock GHz pct 0/50/100", np.percentile(cy / np.maximum(en - st, 1e-9) / 1000, [0, 50, 100]))
