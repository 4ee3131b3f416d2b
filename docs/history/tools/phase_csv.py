#!/usr/bin/env python3
"""Phase breakdown of the waves' first rounds from a STAMP build of k_rows (tools/ubench/k1_anatomy): usage phase_csv.py file.csv"""
import csv, sys
import numpy as np
names = ["0 top -> rows in registers", "1 -> ticket claimed", "2 -> prefetch issued", "3 -> arithmetic done", "4 -> output transposed",
         "5 -> stores issued", "6 -> prefetched units arrived", "7 -> next top (units in LDS)"]
for f in sys.argv[1:]:
    rows = list(csv.DictReader(open(f)))
    nr = np.array([int(r['rounds']) for r in rows])
    P = np.array([[float(x) for x in r['phases'].split()] for r in rows]).reshape(len(rows), 4, 10)
    print(f)
    for r in range(3):
        m = (nr > r + 1) & (P[:, r, 7] > 0) & (P[:, r + 1, 0] > 0)
        if not m.any(): continue
        seg = [P[m, r, i + 1] - P[m, r, i] for i in range(7)] + [P[m, r + 1, 0] - P[m, r, 7]]
        print(" round", r, "waves", m.sum(), "mean total %.2f us" % np.mean(P[m, r + 1, 0] - P[m, r, 0]))
        for n_, s_ in zip(names, seg):
            print("    %-34s mean %.2f  p10 %.2f p50 %.2f p90 %.2f" % (n_, s_.mean(), *np.percentile(s_, [10, 50, 90])))
