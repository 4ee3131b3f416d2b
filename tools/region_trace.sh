#!/bin/bash
# kernel trace of the driver's 20-step command: durations of, and gaps between, the 20 timed launches (the last graph replay before the parity launch)
set -e
export TMPDIR=/tmp
out=gpurun_out/region_trace
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --output-format csv -d $out -- python3 bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-secondary > $out/line.json 2> $out/log.txt
python3 - <<'PY'
import csv, glob
f = sorted(glob.glob("gpurun_out/region_trace/*/*kernel_trace.csv"))[-1]
rows = [r for r in csv.DictReader(open(f)) if "OpProject" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# the timed replay = the last block of 20 launches that are back to back before the final single launches
t = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"])) for r in rows]
# find blocks: a gap > 20 us separates submissions
blocks, cur = [], [t[0]]
for a, b in zip(t, t[1:]):
    if b[0] - a[1] > 20000:
        blocks.append(cur); cur = []
    cur.append(b)
blocks.append(cur)
twenty = [b for b in blocks if len(b) == 20]
print("blocks of 20 back-to-back launches:", len(twenty), "; sizes of the last blocks:", [len(b) for b in blocks[-6:]])
for blk in twenty[-3:]:
    d = [(e - s) / 1e3 for s, e in blk]
    g = [(blk[i + 1][0] - blk[i][1]) / 1e3 for i in range(19)]
    print("durations us:", " ".join("%.1f" % x for x in d))
    print("gaps us     :", " ".join("%.2f" % x for x in g), "| span %.1f us = %.2f per step" % ((blk[-1][1] - blk[0][0]) / 1e3, (blk[-1][1] - blk[0][0]) / 2e4))
PY
